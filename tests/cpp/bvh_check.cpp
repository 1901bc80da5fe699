// Host-logic check of the BVH builder (no GPU): structural invariants of the tree built by rtb::build.
//   1. every primitive index appears in exactly one leaf;
//   2. every child box stored in a node encloses the boxes of all items beneath that child;
//   3. reported depth bounds the real depth.
// usage: bvh_check <n_items> <seed> <leaf_max>   -> prints "OK ..." or "FAIL ..."
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../raytracer-rs_amd/csrc/rt_bvh.h"

static uint32_t rng_state;
static float frand() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)(rng_state >> 8) / 16777216.0f;
}

struct Ctx {
    const std::vector<rtd::Node>& nodes;
    const std::vector<uint32_t>& order;
    const std::vector<rtb::Item>& orig;
    std::vector<int> seen;
    uint32_t max_depth = 0;
    bool ok = true;
};

static void check(Ctx& c, uint32_t ref, const float* lo, const float* hi, uint32_t depth) {
    if (depth > c.max_depth) c.max_depth = depth;
    uint32_t type = ref >> 30;
    if (type == rtd::REF_PRIMS) {
        uint32_t first = ref & 0x03FFFFFFu, cnt = ((ref >> 26) & 15u) + 1u;
        for (uint32_t i = 0; i < cnt; i++) {
            uint32_t pi = c.order[first + i];
            c.seen[pi]++;
            const rtb::Item& it = c.orig[pi];
            for (int k = 0; k < 3; k++)
                if (lo && (it.lo[k] < lo[k] || it.hi[k] > hi[k])) {
                    c.ok = false;
                    std::printf("FAIL prim %u outside its leaf box on axis %d\n", pi, k);
                }
        }
        return;
    }
    if (type != rtd::REF_INNER) return;
    const rtd::Node& n = c.nodes[ref];
    if (lo)
        for (int k = 0; k < 3; k++)
            if (n.lo0[k] < lo[k] || n.hi0[k] > hi[k] || n.lo1[k] < lo[k] || n.hi1[k] > hi[k]) {
                c.ok = false;
                std::printf("FAIL child box outside parent box at node %u\n", ref);
            }
    check(c, n.c0, n.lo0, n.hi0, depth + 1);
    check(c, n.c1, n.lo1, n.hi1, depth + 1);
}

int main(int argc, char** argv) {
    int n = argc > 1 ? std::atoi(argv[1]) : 1000;
    rng_state = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1u;
    uint32_t leaf_max = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 4u;
    std::vector<rtb::Item> items(n);
    for (int i = 0; i < n; i++) {
        float r = 0.05f + 0.45f * frand();
        float c[3] = {-1000.0f + 2000.0f * frand(), 50.0f * frand(), -1000.0f + 2000.0f * frand()};
        if (i == 0) {
            r = 100000.0f;
            c[0] = c[2] = 0.0f;
            c[1] = -100000.0f;
        }
        for (int k = 0; k < 3; k++) {
            items[i].lo[k] = c[k] - r;
            items[i].hi[k] = c[k] + r;
        }
        items[i].payload = (uint32_t)i;
        items[i].type = rtd::REF_PRIMS;
    }
    std::vector<rtb::Item> orig = items;
    std::vector<rtd::Node> nodes;
    std::vector<uint32_t> order;
    rtb::Result r = rtb::build(items, leaf_max, false, 0, nodes, order);
    Ctx c{nodes, order, orig, std::vector<int>(n, 0)};
    check(c, r.root_ref, nullptr, nullptr, 0);
    for (int i = 0; i < n; i++)
        if (c.seen[i] != 1) {
            c.ok = false;
            std::printf("FAIL prim %d appears %d times\n", i, c.seen[i]);
            break;
        }
    if (c.max_depth > r.depth) {
        c.ok = false;
        std::printf("FAIL depth %u > reported %u\n", c.max_depth, r.depth);
    }
    std::printf("%s n=%d nodes=%zu depth=%u\n", c.ok ? "OK" : "FAIL", n, nodes.size(), r.depth);
    return c.ok ? 0 : 1;
}
