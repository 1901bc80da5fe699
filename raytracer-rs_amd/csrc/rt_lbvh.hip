// Linear BVH build on the GPU (gfx950): Morton sort + radix tree + bottom-up box fit.  See rt_lbvh.h.
//
//   k_bounds     box of every primitive -> block reduction -> 12 atomic min/max words (box and centre bounds)
//   k_morton     63-bit Morton code of the box centre (21 bits per axis) + primitive index
//   (rocPRIM)    radix sort of (code, index)
//   k_gather     primitive records into sorted order (leaf i = sorted primitive i)
//   k_leaf_boxes, k_block_boxes, k_table_level   the leaves' boxes and a range structure over them: the union of every
//                aligned block of 64 leaves and, over those blocks, a sparse table (level k: runs of 2^k blocks)
//   k_hierarchy  one thread per inner node: its key range and split by binary search on common prefixes; a node
//                covers a contiguous run of sorted leaves, so the boxes of its two children are two range queries
//   k_depth      one thread per leaf: inner nodes on its path to the root -> atomic max
//   k_widen      one thread per inner node: the four-child form of the finished tree (any builder's)
//
// All of it is HBM-bound streaming work except the sort; nothing here decides a hit (boxes only cull), but the
// box arithmetic is kept identical to the host builder's (-ffp-contract=off) so both builders cull alike.
#include "rt_lbvh.h"

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

namespace rtl {
namespace {

constexpr int BLOCK = 256;
constexpr uint32_t SIDE_BIT = 0x80000000u;
constexpr int MAX_LEVELS = 128;

struct Box3 {
    float lo[3], hi[3];
};

// Box of one primitive record: Sphere / MovingSphere / XY-, XZ-, YZRect::bounding_box (shapes.rs:158-165,
// 236-249, 308-313, 372-377, 436-441), then widened like the host builder does (boxes only cull; the widening
// keeps rounding in the exact primitive tests from placing an accepted hit outside its leaf box).
__device__ Box3 prim_box(const rtd::Prim& p, const rtd::Moving* moving, float t0, float t1) {
    Box3 b;
    const uint32_t kind = p.info & 7u;
    if (kind <= rtd::PK_MOVING_SPHERE) {
        const float r = fabsf(p.a[3]);
        for (int k = 0; k < 3; k++) {
            b.lo[k] = p.a[k] - r;
            b.hi[k] = p.a[k] + r;
        }
        if (kind == rtd::PK_MOVING_SPHERE) {
            const rtd::Moving m = moving[p.aux];
            for (int e = 0; e < 2; e++) {
                const float tt = e ? t1 : t0;
                const float f = (tt - m.t0) / (m.t1 - m.t0);
                for (int k = 0; k < 3; k++) {
                    const float ck = p.a[k] + f * (m.c1[k] - p.a[k]);
                    b.lo[k] = e ? fminf(b.lo[k], ck - r) : ck - r;
                    b.hi[k] = e ? fmaxf(b.hi[k], ck + r) : ck + r;
                }
            }
        }
    } else if (kind == rtd::PK_BOX) {
        // union of the six side rects' boxes: every side is padded by 1e-4 on its thin axis
        const float mn[3] = {p.a[0], p.a[1], p.a[2]}, mx[3] = {p.a[3], p.b0, __uint_as_float(p.aux)};
        for (int k = 0; k < 3; k++) {
            b.lo[k] = fminf(mn[k], mx[k]) - 0.0001f;
            b.hi[k] = fmaxf(mn[k], mx[k]) + 0.0001f;
        }
    } else {
        const int ax = kind == rtd::PK_RECT_XY ? 2 : (kind == rtd::PK_RECT_XZ ? 1 : 0);
        const int a0 = ax == 0 ? 1 : 0, a1 = ax == 2 ? 1 : 2;
        b.lo[ax] = p.a[0] - 0.0001f;
        b.hi[ax] = p.a[0] + 0.0001f;
        b.lo[a0] = fminf(p.a[1], p.a[3]);
        b.hi[a0] = fmaxf(p.a[1], p.a[3]);
        b.lo[a1] = fminf(p.a[2], p.b0);
        b.hi[a1] = fmaxf(p.a[2], p.b0);
    }
    for (int k = 0; k < 3; k++) {
        const float m = fmaxf(fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])), b.hi[k] - b.lo[k]);
        const float pad = 2e-6f * m + 1e-30f;
        b.lo[k] -= pad;
        b.hi[k] += pad;
    }
    return b;
}

// order-preserving map f32 -> u32 (so that integer atomicMin/Max order floats)
__device__ uint32_t ordered(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unordered(uint32_t u) {
    const uint32_t v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    union {
        uint32_t i;
        float f;
    } c;
    c.i = v;
    return c.f;
}

// scalars[0..2] min of box lo, [3..5] max of box hi, [6..8] min of centre, [9..11] max of centre, [12] depth
__global__ __launch_bounds__(BLOCK) void k_bounds(const rtd::Prim* prims, const rtd::Moving* moving, uint32_t n, float t0, float t1,
                                                   uint32_t* scalars) {
    __shared__ uint32_t red[12];
    if (threadIdx.x < 12) red[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    uint32_t v[12];
    for (int k = 0; k < 12; k++) v[k] = (k % 6) < 3 ? 0xFFFFFFFFu : 0u;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const Box3 b = prim_box(prims[i], moving, t0, t1);
        for (int k = 0; k < 3; k++) {
            const float c = 0.5f * (b.lo[k] + b.hi[k]);
            v[k] = min(v[k], ordered(b.lo[k]));
            v[3 + k] = max(v[3 + k], ordered(b.hi[k]));
            v[6 + k] = min(v[6 + k], ordered(c));
            v[9 + k] = max(v[9 + k], ordered(c));
        }
    }
    for (int k = 0; k < 12; k++) {
        if ((k % 6) < 3)
            atomicMin(&red[k], v[k]);
        else
            atomicMax(&red[k], v[k]);
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        if ((threadIdx.x % 6) < 3)
            atomicMin(&scalars[threadIdx.x], red[threadIdx.x]);
        else
            atomicMax(&scalars[threadIdx.x], red[threadIdx.x]);
    }
}

__device__ unsigned long long spread21(uint32_t x) {  // bit i of x -> bit 3i
    unsigned long long v = x & 0x1FFFFFu;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ __launch_bounds__(BLOCK) void k_morton(const rtd::Prim* prims, const rtd::Moving* moving, uint32_t n, float t0, float t1,
                                                   const uint32_t* scalars, unsigned long long* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const Box3 b = prim_box(prims[i], moving, t0, t1);
    // one cell size for all three axes (the largest extent of the centres spans the 2^21 cells): a flat or
    // elongated scene is then not split along its thin axis before the cells have become roughly cubic
    float ext = 0.0f;
    for (int k = 0; k < 3; k++) ext = fmaxf(ext, unordered(scalars[9 + k]) - unordered(scalars[6 + k]));
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = unordered(scalars[6 + k]);
        const float c = 0.5f * (b.lo[k] + b.hi[k]);
        float f = ext > 0.0f ? (c - lo) / ext : 0.0f;
        f = fminf(fmaxf(f, 0.0f), 1.0f);
        const uint32_t cell = (uint32_t)(f * 2097152.0f);
        q[k] = cell > 2097151u ? 2097151u : cell;
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    vals[i] = i;
}

__global__ __launch_bounds__(BLOCK) void k_gather(const rtd::Prim* unsorted, const uint32_t* order, uint32_t n, rtd::Prim* sorted) {
    // two lanes per 32-byte record: 16-byte loads and stores
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t i = t >> 1, half = t & 1u;
    if (i >= n) return;
    const float4* src = reinterpret_cast<const float4*>(unsorted + order[i]);
    reinterpret_cast<float4*>(sorted + i)[half] = src[half];
}

// length of the common prefix of keys i and j (ties broken by the index, so that all keys are distinct); -1 outside
__device__ int prefix(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// ---- boxes by range query -----------------------------------------------------------------------------------
// Every inner node of the radix tree covers a contiguous run [first, last] of the sorted leaves, and min / max are exact
// and associative: a child's box is the union of the leaf boxes of its run, whichever way the union is bracketed.  So
// nothing has to climb the tree (one thread per leaf meeting its sibling at every level costs two device-wide fences
// per node: 80 ms for 10 M leaves); a node reads its boxes from a structure made in three streaming passes: the leaf
// boxes, the union of each aligned block of FIT_BLOCK leaves, and a sparse table over the blocks.  A run shorter than
// two blocks is read leaf by leaf, a longer one as two partial blocks plus two table entries.
constexpr uint32_t FIT_BLOCK = 64;

__device__ Box3 load_box(const float4* boxes, uint32_t i) {
    const float4 lo = boxes[2u * i], hi = boxes[2u * i + 1u];
    return Box3{{lo.x, lo.y, lo.z}, {hi.x, hi.y, hi.z}};
}
__device__ void merge_box(Box3& b, const float4* boxes, uint32_t i) {
    const float4 lo = boxes[2u * i], hi = boxes[2u * i + 1u];
    b.lo[0] = fminf(b.lo[0], lo.x);
    b.lo[1] = fminf(b.lo[1], lo.y);
    b.lo[2] = fminf(b.lo[2], lo.z);
    b.hi[0] = fmaxf(b.hi[0], hi.x);
    b.hi[1] = fmaxf(b.hi[1], hi.y);
    b.hi[2] = fmaxf(b.hi[2], hi.z);
}
__device__ void store_box(float4* boxes, uint32_t i, const Box3& b) {
    boxes[2u * i] = make_float4(b.lo[0], b.lo[1], b.lo[2], 0.0f);
    boxes[2u * i + 1u] = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.0f);
}
__global__ __launch_bounds__(BLOCK) void k_leaf_boxes(const rtd::Prim* sorted, const rtd::Moving* moving, uint32_t n, float t0, float t1,
                                                       float4* leaf) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) store_box(leaf, i, prim_box(sorted[i], moving, t0, t1));
}
// level 0 of the table: block b = leaves [b * FIT_BLOCK, (b + 1) * FIT_BLOCK) (the last block may be short)
__global__ __launch_bounds__(BLOCK) void k_block_boxes(const float4* leaf, uint32_t n, uint32_t nb, float4* table) {
    const uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    if (b >= nb) return;
    const uint32_t first = b * FIT_BLOCK, end = min(n, first + FIT_BLOCK);
    Box3 box = load_box(leaf, first);
    for (uint32_t i = first + 1u; i < end; i++) merge_box(box, leaf, i);
    store_box(table, b, box);
}
// level k (runs of 2^k blocks starting at b) from level k - 1
__global__ __launch_bounds__(BLOCK) void k_table_level(float4* table, uint32_t nb, uint32_t k) {
    const uint32_t b = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t half = 1u << (k - 1u);
    if (b + 2u * half > nb) return;
    const float4* below = table + 2ull * (k - 1u) * nb;
    Box3 box = load_box(below, b);
    merge_box(box, below, b + half);
    store_box(table + 2ull * k * nb, b, box);
}
// union of the boxes of leaves l .. r (inclusive)
__device__ Box3 range_box(const float4* leaf, const float4* table, uint32_t nb, uint32_t l, uint32_t r) {
    Box3 b = load_box(leaf, l);
    if (r - l < 2u * FIT_BLOCK) {
        for (uint32_t i = l + 1u; i <= r; i++) merge_box(b, leaf, i);
        return b;
    }
    // r - l >= 2 * FIT_BLOCK: at least one whole block lies strictly between the blocks of l and r
    const uint32_t bl = l / FIT_BLOCK + 1u, br = r / FIT_BLOCK - 1u;
    for (uint32_t i = l + 1u; i < bl * FIT_BLOCK; i++) merge_box(b, leaf, i);
    for (uint32_t i = (br + 1u) * FIT_BLOCK; i <= r; i++) merge_box(b, leaf, i);
    const uint32_t k = 31u - (uint32_t)__clz((int)(br - bl + 1u));
    merge_box(b, table + 2ull * k * nb, bl);
    merge_box(b, table + 2ull * k * nb, br + 1u - (1u << k));
    return b;
}

// Radix tree over the sorted keys: inner node i covers a key range one of whose ends is i.
__global__ __launch_bounds__(BLOCK) void k_hierarchy(const unsigned long long* keys, uint32_t n, uint32_t prim_base, uint32_t node_base,
                                                      rtd::Node* nodes, uint32_t* parent_inner, uint32_t* parent_leaf, const float4* leaf,
                                                      const float4* table, uint32_t nb) {
    const int i = (int)(blockIdx.x * BLOCK + threadIdx.x);
    const int nn = (int)n;
    if (i >= nn - 1) return;
    const int d = prefix(keys, nn, i, i + 1) - prefix(keys, nn, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, nn, i, i - d);
    int lmax = 2;
    while (prefix(keys, nn, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, nn, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix(keys, nn, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix(keys, nn, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int split = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const bool leaf0 = split == first, leaf1 = split + 1 == last;
    rtd::Node& nd = nodes[i];
    // leaf: REF_PRIMS, one primitive; inner: REF_INNER
    nd.c0 = leaf0 ? ((rtd::REF_PRIMS << 30) | (prim_base + (uint32_t)split)) : (node_base + (uint32_t)split);
    nd.c1 = leaf1 ? ((rtd::REF_PRIMS << 30) | (prim_base + (uint32_t)split + 1u)) : (node_base + (uint32_t)split + 1u);
    nd.pad[0] = nd.pad[1] = 0;
    if (leaf0)
        parent_leaf[split] = (uint32_t)i;
    else
        parent_inner[split] = (uint32_t)i;
    if (leaf1)
        parent_leaf[split + 1] = (uint32_t)i | SIDE_BIT;
    else
        parent_inner[split + 1] = (uint32_t)i | SIDE_BIT;
    const Box3 b0 = range_box(leaf, table, nb, (uint32_t)first, (uint32_t)split);
    const Box3 b1 = range_box(leaf, table, nb, (uint32_t)split + 1u, (uint32_t)last);
    for (int k = 0; k < 3; k++) {
        nd.lo0[k] = b0.lo[k];
        nd.hi0[k] = b0.hi[k];
        nd.lo1[k] = b1.lo[k];
        nd.hi1[k] = b1.hi[k];
    }
}

__global__ __launch_bounds__(BLOCK) void k_depth(uint32_t n, const uint32_t* parent_inner, const uint32_t* parent_leaf, uint32_t* depth) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    uint32_t d = 0;
    if (i < n) {
        uint32_t pi = parent_leaf[i] & ~SIDE_BIT;
        d = 1;
        while (pi != 0u && d < (uint32_t)MAX_LEVELS) {
            pi = parent_inner[pi] & ~SIDE_BIT;
            d++;
        }
    }
    for (int off = 32; off > 0; off >>= 1) d = max(d, (uint32_t)__shfl_down((int)d, off, 64));
    if ((threadIdx.x & 63u) == 0 && d) atomicMax(depth, d);
}

// Four-child form of binary node i: the children of its inner children, or the child itself where that is a
// leaf, an instance or empty.  One thread per node; a map over the array (the entries of odd tree depth are
// never reached by a walk from the root, they are written all the same).
__global__ __launch_bounds__(BLOCK) void k_widen(uint32_t n, const rtd::Node* nodes, rtd::WideNode* wide) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const rtd::Node b = nodes[i];
    rtd::WideNode out;
    for (int k = 0; k < 2; k++) {
        const uint32_t ref = k ? b.c1 : b.c0;
        const float* lo = k ? b.lo1 : b.lo0;
        const float* hi = k ? b.hi1 : b.hi0;
        int s = 2 * k;
        if (ref != rtd::REF_EMPTY && (ref >> 30) == rtd::REF_INNER && ref < n) {
            const rtd::Node g = nodes[ref];
            for (int a = 0; a < 3; a++) {
                out.box[s][a] = g.lo0[a];
                out.box[s][3 + a] = g.hi0[a];
                out.box[s + 1][a] = g.lo1[a];
                out.box[s + 1][3 + a] = g.hi1[a];
            }
            out.c[s] = g.c0;
            out.c[s + 1] = g.c1;
        } else {
            for (int a = 0; a < 3; a++) {
                out.box[s][a] = lo[a];
                out.box[s][3 + a] = hi[a];
                out.box[s + 1][a] = 0.0f;
                out.box[s + 1][3 + a] = 0.0f;
            }
            out.c[s] = ref;
            out.c[s + 1] = rtd::REF_EMPTY;
        }
    }
    for (int a = 0; a < 4; a++) out.pad[a] = 0u;
    wide[i] = out;
}

hipError_t dev_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 16); }
inline int grid_for(uint64_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

}  // namespace

void Workspace::release() {
    void* all[] = {keys[0], keys[1], vals[0], vals[1], sort_tmp, parent_inner, parent_leaf, leaf_boxes, box_table, scalars};
    for (void* p : all)
        if (p) (void)hipFree(p);
    *this = Workspace();
}

#define RTL_TRY(call)                       \
    do {                                    \
        hipError_t e_ = (call);             \
        if (e_ != hipSuccess) return e_;    \
    } while (0)

hipError_t sort_world(hipStream_t s, Workspace& w, const rtd::Prim* unsorted, const rtd::Moving* moving, uint32_t n,
                      float shutter_t0, float shutter_t1, rtd::Prim* sorted, float lo[3], float hi[3]) {
    if (n < 2) return hipErrorInvalidValue;
    w.release();
    w.n = n;
    for (int k = 0; k < 2; k++) {
        RTL_TRY(dev_alloc(&w.keys[k], (size_t)n * sizeof(unsigned long long)));
        RTL_TRY(dev_alloc(&w.vals[k], (size_t)n * sizeof(uint32_t)));
    }
    RTL_TRY(dev_alloc((void**)&w.parent_inner, (size_t)n * sizeof(uint32_t)));
    RTL_TRY(dev_alloc((void**)&w.parent_leaf, (size_t)n * sizeof(uint32_t)));
    RTL_TRY(dev_alloc((void**)&w.scalars, 16 * sizeof(uint32_t)));
    uint32_t init[16];
    for (int k = 0; k < 12; k++) init[k] = (k % 6) < 3 ? 0xFFFFFFFFu : 0u;
    for (int k = 12; k < 16; k++) init[k] = 0u;
    RTL_TRY(hipMemcpyAsync(w.scalars, init, sizeof(init), hipMemcpyHostToDevice, s));
    RTL_TRY(hipStreamSynchronize(s));  // init lives on this stack frame

    const int red_grid = (int)std::min<uint64_t>(grid_for(n), 2048);
    hipLaunchKernelGGL(k_bounds, dim3(red_grid), dim3(BLOCK), 0, s, unsorted, moving, n, shutter_t0, shutter_t1, w.scalars);
    RTL_TRY(hipGetLastError());
    auto* keys_in = (unsigned long long*)w.keys[0];
    auto* keys_out = (unsigned long long*)w.keys[1];
    auto* vals_in = (uint32_t*)w.vals[0];
    auto* vals_out = (uint32_t*)w.vals[1];
    hipLaunchKernelGGL(k_morton, dim3(grid_for(n)), dim3(BLOCK), 0, s, unsorted, moving, n, shutter_t0, shutter_t1, w.scalars, keys_in, vals_in);
    RTL_TRY(hipGetLastError());
    size_t tmp_bytes = 0;
    RTL_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 63u, s));
    RTL_TRY(dev_alloc(&w.sort_tmp, tmp_bytes));
    w.sort_tmp_bytes = tmp_bytes;
    RTL_TRY(rocprim::radix_sort_pairs(w.sort_tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 63u, s));
    hipLaunchKernelGGL(k_gather, dim3(grid_for(2ull * n)), dim3(BLOCK), 0, s, unsorted, vals_out, n, sorted);
    RTL_TRY(hipGetLastError());
    uint32_t back[12];
    RTL_TRY(hipMemcpyAsync(back, w.scalars, sizeof(back), hipMemcpyDeviceToHost, s));
    RTL_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 3; k++) {
        lo[k] = unordered(back[k]);
        hi[k] = unordered(back[3 + k]);
    }
    return hipSuccess;
}

hipError_t build_tree(hipStream_t s, Workspace& w, const rtd::Prim* sorted, const rtd::Moving* moving, uint32_t n,
                      float shutter_t0, float shutter_t1, uint32_t prim_base, rtd::Node* nodes, uint32_t node_base,
                      uint32_t* depth) {
    if (n < 2 || w.n != n || !w.keys[1]) return hipErrorInvalidValue;
    const auto* keys = (const unsigned long long*)w.keys[1];  // sorted
    // range structure over the leaf boxes (see range_box)
    const uint32_t nb = (n + FIT_BLOCK - 1u) / FIT_BLOCK;
    uint32_t levels = 1;
    while ((2u << (levels - 1u)) <= nb) levels++;
    if (!w.leaf_boxes) RTL_TRY(dev_alloc((void**)&w.leaf_boxes, 2ull * n * sizeof(float4)));
    if (!w.box_table) RTL_TRY(dev_alloc((void**)&w.box_table, 2ull * levels * nb * sizeof(float4)));
    float4* leaf = (float4*)w.leaf_boxes;
    float4* table = (float4*)w.box_table;
    hipLaunchKernelGGL(k_leaf_boxes, dim3(grid_for(n)), dim3(BLOCK), 0, s, sorted, moving, n, shutter_t0, shutter_t1, leaf);
    RTL_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_block_boxes, dim3(grid_for(nb)), dim3(BLOCK), 0, s, leaf, n, nb, table);
    RTL_TRY(hipGetLastError());
    for (uint32_t k = 1; k < levels; k++) {
        hipLaunchKernelGGL(k_table_level, dim3(grid_for(nb)), dim3(BLOCK), 0, s, table, nb, k);
        RTL_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_hierarchy, dim3(grid_for(n - 1)), dim3(BLOCK), 0, s, keys, n, prim_base, node_base, nodes, w.parent_inner,
                       w.parent_leaf, leaf, table, nb);
    RTL_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_depth, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, w.parent_inner, w.parent_leaf, w.scalars + 12);
    RTL_TRY(hipGetLastError());
    uint32_t d = 0;
    RTL_TRY(hipMemcpyAsync(&d, w.scalars + 12, sizeof(d), hipMemcpyDeviceToHost, s));
    RTL_TRY(hipStreamSynchronize(s));
    *depth = d;
    return hipSuccess;
}

hipError_t widen_tree(hipStream_t s, const rtd::Node* nodes, uint32_t n, rtd::WideNode* wide) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_widen, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, nodes, wide);
    return hipGetLastError();
}

}  // namespace rtl
