// Host-side builder of the linear BVH the kernels traverse.
//
// Stands in for BVHNode::from_vector (bvh_node.rs:52-120).  The reference builds a pointer tree with
// a random split axis and a clone-and-sort per level; any valid BVH yields the same closest hits, so
// this builder uses a binned surface-area heuristic (fewer node visits per ray), emits 64-byte nodes
// that hold both child boxes, and orders primitives by leaf so that a leaf is one contiguous range.
#pragma once
#include <stdint.h>

#include <vector>

#include "rt_device.h"

namespace rtb {

struct Item {
    float lo[3], hi[3];
    uint32_t payload;  // primitive: index into the caller's primitive array; group: group index
    uint32_t type;     // rtd::REF_PRIMS or rtd::REF_INSTANCE
};

struct Result {
    uint32_t root_ref = rtd::REF_EMPTY;
    uint32_t depth = 0;  // number of inner levels on the longest root-to-leaf path
};

// Builds a BVH over items[0..n).  Inner nodes are appended to `nodes`; primitive payloads are
// appended to `prim_order` in leaf order, and leaf references index prim_base + position in that
// appended range.  leaf_max <= rtd::MAX_LEAF_PRIMS.  single_leaf: put all items (primitives only,
// n <= MAX_LEAF_PRIMS) into one leaf in the given order.
Result build(std::vector<Item>& items, uint32_t leaf_max, bool single_leaf, uint32_t prim_base,
             std::vector<rtd::Node>& nodes, std::vector<uint32_t>& prim_order);

}  // namespace rtb
