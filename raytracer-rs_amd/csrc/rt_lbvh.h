// Device-side builder of the world-level BVH (linear BVH over Morton-sorted primitives).
//
// Stands in for BVHNode::from_vector (bvh_node.rs:52-120) on scenes whose world holds many primitives:
// the reference clones and sorts the object vector on every level (O(n log^2 n) on one thread); here the
// primitives are sorted once by the Morton code of their box centre (rocPRIM radix sort), the radix tree
// over the sorted codes is emitted with one thread per inner node, and the boxes are fitted bottom-up
// with one thread per leaf.  Any valid BVH yields the same closest hits, so the result is exchangeable
// with the host builder's (rt_bvh.h); only the number of node visits per ray differs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace rtl {

// Temporary device memory of one build; released by release().
struct Workspace {
    void* keys[2] = {nullptr, nullptr};   // 64-bit Morton codes (radix sort double buffer)
    void* vals[2] = {nullptr, nullptr};   // primitive index before sorting
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint32_t* parent_inner = nullptr;     // n-1 entries: parent node | child slot << 31
    uint32_t* parent_leaf = nullptr;      // n entries
    void* leaf_boxes = nullptr;           // n boxes (two float4 each) of the sorted primitives
    void* box_table = nullptr;            // unions of aligned blocks of 64 leaves and a sparse table over the blocks
    uint32_t* scalars = nullptr;          // 12 ordered-int bounds words + depth word
    uint32_t n = 0;
    void release();
};

// Phase 1: boxes of the n world primitives (same formulas and widening as the host builder), their
// bounds, Morton codes of the box centres, sort; writes the primitives in sorted order to `sorted`
// (device, n records) and the union of all boxes to lo/hi (host).  `unsorted` and `moving` are device
// arrays; shutter_t0/t1 bound the motion of moving spheres (MovingSphere::bounding_box, shapes.rs:236-249).
hipError_t sort_world(hipStream_t s, Workspace& w, const rtd::Prim* unsorted, const rtd::Moving* moving, uint32_t n,
                      float shutter_t0, float shutter_t1, rtd::Prim* sorted, float lo[3], float hi[3]);

// Phase 2: the n-1 inner nodes over the sorted primitives, written to nodes[0 .. n-1); inner node i has
// the global index node_base + i (the root is node_base), leaf i references primitive prim_base + i.
// depth = inner nodes on the longest root-to-leaf path.  Requires n >= 2.
hipError_t build_tree(hipStream_t s, Workspace& w, const rtd::Prim* sorted, const rtd::Moving* moving, uint32_t n,
                      float shutter_t0, float shutter_t1, uint32_t prim_base, rtd::Node* nodes, uint32_t node_base,
                      uint32_t* depth);

// Four-child form of a finished tree of n binary nodes (host- or device-built, instances' subtrees included):
// wide[i] holds the grandchildren of nodes[i].  Both arrays on the device; asynchronous on s.
hipError_t widen_tree(hipStream_t s, const rtd::Node* nodes, uint32_t n, rtd::WideNode* wide);

}  // namespace rtl
