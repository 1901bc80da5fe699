#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rtb {
namespace {

constexpr int NBINS = 16;
constexpr uint32_t SOFT_DEPTH = 44;  // below this depth: SAH splits; from here on: balanced median splits

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float* l, const float* h) {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], l[k]);
            hi[k] = std::max(hi[k], h[k]);
        }
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Builder {
    std::vector<Item>& items;
    uint32_t leaf_max;
    uint32_t prim_base;
    std::vector<rtd::Node>& nodes;
    std::vector<uint32_t>& order;
    size_t order_start;
    uint32_t max_depth = 0;

    uint32_t make_leaf(size_t b, size_t e) {
        uint32_t first = prim_base + (uint32_t)(order.size() - order_start);
        for (size_t i = b; i < e; i++) order.push_back(items[i].payload);
        return rtd::make_prim_ref(first, (uint32_t)(e - b));
    }

    // returns child reference; box of the range in `out`
    uint32_t build_range(size_t b, size_t e, uint32_t depth, Box& out) {
        const size_t n = e - b;
        out.reset();
        bool all_prims = true;
        Box cb;  // centroid bounds
        cb.reset();
        for (size_t i = b; i < e; i++) {
            out.grow(items[i].lo, items[i].hi);
            float c[3];
            for (int k = 0; k < 3; k++) c[k] = 0.5f * (items[i].lo[k] + items[i].hi[k]);
            cb.grow(c, c);
            if (items[i].type != rtd::REF_PRIMS) all_prims = false;
        }
        max_depth = std::max(max_depth, depth);
        if (n == 1) {
            if (items[b].type == rtd::REF_PRIMS) return make_leaf(b, e);
            return rtd::make_ref(items[b].type, items[b].payload);
        }
        const bool leaf_ok = all_prims && n <= leaf_max;
        size_t mid = b;
        bool have_split = false;
        if (depth < SOFT_DEPTH) {
            // binned SAH over the three axes
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            const float parent_area = std::max(out.half_area(), 1e-30f);
            for (int ax = 0; ax < 3; ax++) {
                float ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.0f)) continue;
                Box bins[NBINS];
                uint32_t cnt[NBINS];
                for (int i = 0; i < NBINS; i++) {
                    bins[i].reset();
                    cnt[i] = 0;
                }
                const float scale = (float)NBINS / ext;
                for (size_t i = b; i < e; i++) {
                    float c = 0.5f * (items[i].lo[ax] + items[i].hi[ax]);
                    int bi = (int)((c - cb.lo[ax]) * scale);
                    bi = std::min(std::max(bi, 0), NBINS - 1);
                    bins[bi].grow(items[i].lo, items[i].hi);
                    cnt[bi]++;
                }
                float right_area[NBINS];
                uint32_t right_cnt[NBINS];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int i = NBINS - 1; i > 0; i--) {
                    acc.grow(bins[i]);
                    c += cnt[i];
                    right_area[i] = acc.half_area();
                    right_cnt[i] = c;
                }
                acc.reset();
                c = 0;
                for (int i = 0; i < NBINS - 1; i++) {
                    acc.grow(bins[i]);
                    c += cnt[i];
                    if (c == 0 || right_cnt[i + 1] == 0) continue;
                    float cost = 1.0f + (acc.half_area() * (float)c + right_area[i + 1] * (float)right_cnt[i + 1]) / parent_area;
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = ax;
                        best_bin = i;
                    }
                }
            }
            if (best_axis >= 0) {
                if (leaf_ok && best_cost >= (float)n) return make_leaf(b, e);
                const float ext = cb.hi[best_axis] - cb.lo[best_axis];
                const float scale = (float)NBINS / ext;
                const float lo = cb.lo[best_axis];
                auto it = std::partition(items.begin() + b, items.begin() + e, [&](const Item& it_) {
                    float c = 0.5f * (it_.lo[best_axis] + it_.hi[best_axis]);
                    int bi = (int)((c - lo) * scale);
                    bi = std::min(std::max(bi, 0), NBINS - 1);
                    return bi <= best_bin;
                });
                mid = (size_t)(it - items.begin());
                have_split = mid > b && mid < e;
            }
        }
        if (!have_split) {
            if (leaf_ok) return make_leaf(b, e);
            // median split on the widest centroid axis (also the depth-bounded fallback)
            int ax = 0;
            float best = -1.0f;
            for (int k = 0; k < 3; k++) {
                float ext = cb.hi[k] - cb.lo[k];
                if (ext > best) {
                    best = ext;
                    ax = k;
                }
            }
            mid = b + n / 2;
            std::nth_element(items.begin() + b, items.begin() + mid, items.begin() + e, [ax](const Item& x, const Item& y) {
                return (x.lo[ax] + x.hi[ax]) < (y.lo[ax] + y.hi[ax]);
            });
        }
        const uint32_t ni = (uint32_t)nodes.size();
        nodes.emplace_back();
        Box b0, b1;
        uint32_t c0 = build_range(b, mid, depth + 1, b0);
        uint32_t c1 = build_range(mid, e, depth + 1, b1);
        rtd::Node& nd = nodes[ni];
        for (int k = 0; k < 3; k++) {
            nd.lo0[k] = b0.lo[k];
            nd.hi0[k] = b0.hi[k];
            nd.lo1[k] = b1.lo[k];
            nd.hi1[k] = b1.hi[k];
        }
        nd.c0 = c0;
        nd.c1 = c1;
        nd.pad[0] = nd.pad[1] = 0;
        return rtd::make_ref(rtd::REF_INNER, ni);
    }
};

}  // namespace

Result build(std::vector<Item>& items, uint32_t leaf_max, bool single_leaf, uint32_t prim_base,
             std::vector<rtd::Node>& nodes, std::vector<uint32_t>& prim_order) {
    Result r;
    if (items.empty()) return r;
    leaf_max = std::max(1u, std::min(leaf_max, rtd::MAX_LEAF_PRIMS));
    Builder B{items, leaf_max, prim_base, nodes, prim_order, prim_order.size()};
    if (single_leaf) {
        r.root_ref = B.make_leaf(0, items.size());
        r.depth = 0;
        return r;
    }
    Box box;
    r.root_ref = B.build_range(0, items.size(), 0, box);
    r.depth = B.max_depth;
    return r;
}

}  // namespace rtb
