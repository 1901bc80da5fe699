// Device-side data layout shared by the host builder and the HIP kernels.
//
// Everything the kernels read lives in HBM as flat arrays:
//   prims[]   32-B records in BVH leaf order (sphere: centre+radius, rect: plane+bounds),
//   moving[]  32-B side records for moving spheres (second centre, shutter times),
//   nodes[]   64-B inner nodes holding BOTH child boxes and both child references,
//   wide[]    optional 128-B four-child form of the same nodes,
//   groups[]  instance / medium descriptors (3x4 transforms, medium density, root reference),
//   mats[], texs[], perlin[], images[].
#pragma once
#include <stdint.h>

namespace rtd {

// ---- child / root references (32 bit) ---------------------------------------
// bits 31..30 type, for REF_PRIMS bits 29..26 = count-1, bits 25..0 = first prim
constexpr uint32_t REF_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t REF_SENTINEL = 0xFFFFFFFEu;  // traversal stack marker: leave instance space
constexpr uint32_t REF_INNER = 0u, REF_PRIMS = 1u, REF_INSTANCE = 2u;
constexpr uint32_t MAX_LEAF_PRIMS = 16u;
constexpr uint32_t MAX_PRIMS = 1u << 26;
inline uint32_t make_ref(uint32_t type, uint32_t index) { return (type << 30) | index; }
inline uint32_t make_prim_ref(uint32_t first, uint32_t count) { return (REF_PRIMS << 30) | ((count - 1u) << 26) | first; }

// The crown: copies of the inner nodes a ray is most likely to visit (the top of the world's tree and of the instances'
// trees, picked at commit by box area), which the persistent trace kernel keeps in LDS.  Inside those copies a child
// that is itself in the crown is referred to as an inner node with index CROWN_BASE + its place in the crown: no tree
// has that many nodes, so the walk tells the two apart with one compare.  Only walks that start at Scene::crown_root
// ever meet such a reference.
constexpr uint32_t CROWN_BASE = 0x3FFF0000u, CROWN_MAX = 0xFFFFu;
constexpr uint32_t CROWN_INST_SLOTS = 64u;  // instance groups below this index can have their root in the crown

constexpr uint32_t HIT_NONE = 0xFFFFFFFFu;
constexpr uint32_t HIT_MEDIUM_BIT = 0x80000000u;  // hit.prim = HIT_MEDIUM_BIT | group index
constexpr uint32_t MEDIUM_KEY_BASE = 0x40000000u; // tie key of the k-th medium

// PK_BOX: the six rects of a Box2 (shapes.rs:456-466) kept as ONE record and one BVH leaf.  The kernels still
// evaluate the six rect tests in the reference's side order with the reference's arithmetic, so hits, ids and
// ties are those of six separate rects; what changes is the tree (one leaf instead of six).
enum PrimKind : uint32_t { PK_SPHERE = 0, PK_MOVING_SPHERE = 1, PK_RECT_XY = 2, PK_RECT_XZ = 3, PK_RECT_YZ = 4, PK_BOX = 5 };

struct alignas(32) Prim {
    float a[4];     // sphere: cx,cy,cz,r        rect: k, min0, min1, max0       box: min.x, min.y, min.z, max.x
    float b0;       // sphere: unused            rect: max1                      box: max.y
    uint32_t info;  // bits 2..0 PrimKind, bits 31..8 material index
    uint32_t id;    // caller's flat id (tie key, reported by rt_intersect_batch); box: id of its first side
    uint32_t aux;   // moving sphere: index into moving[]                        box: bits of max.z
};
static_assert(sizeof(Prim) == 32, "Prim must be 32 B");

struct alignas(32) Moving {
    float c1[3];
    float t0, t1;
    float pad[3];
};
static_assert(sizeof(Moving) == 32, "Moving must be 32 B");

struct alignas(64) Node {
    float lo0[3], hi0[3];  // box of child 0
    float lo1[3], hi1[3];  // box of child 1
    uint32_t c0, c1;
    uint32_t pad[2];
};
static_assert(sizeof(Node) == 64, "Node must be 64 B");

// Four-wide form of the same tree: wide[i] holds the GRANDchildren of binary node i (a child that is not an
// inner node keeps its slot, its partner slot is empty).  A walk that starts at the root only ever lands on
// binary nodes of even depth, so the entries of odd depth are never read; keeping one entry per binary node
// makes the conversion a map over the node array (no compaction, same indices, same references).
struct alignas(128) WideNode {
    float box[4][6];  // slot s: lo.xyz, hi.xyz
    uint32_t c[4];    // REF_EMPTY: unused slot
    uint32_t pad[4];
};
static_assert(sizeof(WideNode) == 128, "WideNode must be 128 B");

struct alignas(16) Group {
    float m[12];     // object -> world, rows
    float minv[12];  // world -> object
    double neg_inv_density;
    uint32_t root_ref;  // instance: BLAS root; medium: first boundary primitive in bits 25..0 (media are not BVH leaves; count in depth)
    uint32_t mat;
    uint32_t kind;      // 1 instance, 2 medium
    uint32_t has_xform;
    uint32_t order;     // instance: instance index; medium: medium index (RNG salt, reported id)
    uint32_t flat_range;  // the group's primitives as first | (count-1) << 26, FLAT_NONE when empty or more than 64
    float blo[3], bhi[3]; // widened box of the group's primitives in the group's own space (culls medium queries)
    // An instance may sit inside another instance (TransformedObject<T> takes any Hittable, shapes.rs:26-41): parent
    // is the enclosing instance group or -1, depth the number of instances around this one (at most MAX_NESTING - 1).
    int32_t parent;
    uint32_t depth;  // (a medium: the number of primitives of its boundary)
};
constexpr uint32_t MAX_NESTING = 4u;
constexpr uint32_t FLAT_NONE = 0xFFFFFFFFu;
static_assert(sizeof(Group) == 160, "Group must be 160 B");

enum MatKind : uint32_t { MK_LAMBERTIAN = 0, MK_METAL = 1, MK_DIELECTRIC = 2, MK_LIGHT = 3, MK_ISOTROPIC = 4 };
// flags the commit step writes into Material::pad
constexpr uint32_t MF_NEEDS_UV = 1u;  // the material's texture (or a checker child of it) is an image
constexpr uint32_t MF_SOLID = 2u;     // the texture is a SolidColor; its colour has been copied into albedo
struct alignas(16) Material {
    uint32_t kind, tex;
    float albedo[3];  // Metal: albedo; with MF_SOLID: the solid texture's colour
    float fuzz, ior;
    uint32_t pad;     // MF_* flags
};
static_assert(sizeof(Material) == 32, "Material must be 32 B");

enum TexKind : uint32_t { TK_SOLID = 0, TK_CHECKER = 1, TK_NOISE = 2, TK_IMAGE = 3 };
struct alignas(16) Texture {
    uint32_t kind;
    float color[3];
    uint32_t odd, even;
    float freq;
    uint32_t table;
};
static_assert(sizeof(Texture) == 32, "Texture must be 32 B");

struct alignas(16) PerlinTable {
    float vec[256][4];
    int32_t perm[3][256];
};

struct Image {
    const uint8_t* rgb;
    uint32_t w, h;
};

struct Camera {
    float origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius, time_start, time_end;
};

struct Scene {
    const Prim* prims;
    const Moving* moving;
    const Node* nodes;
    const WideNode* wide;  // non-null: inner nodes are walked four children at a time
    const Group* groups;
    const Material* mats;
    const Texture* texs;
    const PerlinTable* perlin;
    const Image* images;
    const uint32_t* media;  // group indices of the media, in creation order (evaluated after the BVH walk)
    uint32_t root_ref;
    uint32_t n_prims, n_groups, n_media;
    // Small scenes skip the tree: every ray tests the world primitives [world_first, world_first + world_count)
    // and each instance's flat_range in a loop that is uniform over the wave (flat != 0).
    uint32_t flat, world_first, world_count;
    uint32_t stack_cap;  // entries of the per-lane traversal stack the kernels are launched with (LDS, dynamic)
    uint32_t n_perlin;   // Perlin tables; the shading kernels keep table 0 in LDS when there is one
    // lds_scene != 0: the scene is small enough that the shading kernels also keep the primitive, material and
    // texture records in LDS (n_prims, n_mats, n_texs records of 32 bytes each)
    uint32_t lds_scene, n_mats, n_texs;
    uint32_t nested;  // != 0: some instance stands inside another one (Group::parent >= 0 somewhere)
    // The surfaces that can be met inside the first medium: indices (into prims[]) of the world primitives whose box meets
    // that medium's boundary, stored behind media[].  NEAR_NONE: not known (an instance reaches into the boundary's box,
    // the boundary is transformed, the world's tree was built on the device, more than 32 such primitives).
    uint32_t near0_n;
    uint32_t near0_ball;  // != 0: that boundary is one resting sphere, and the list was made for the ball 1.001 times its radius
    // != 0 (needs near0_ball): every medium (at most 4) is one resting sphere in world space with a plain-colour phase
    // material, and the near list holds at most one primitive, a resting sphere: k_settled may take its specialised step
    uint32_t settle_fast;
    // The crown (see CROWN_BASE), or crown_n == 0.  Device memory: CROWN_INST_SLOTS words (word g: the crown reference of
    // instance group g's root, REF_EMPTY when it is not in the crown), then crown_n records (Node, or WideNode when wide is
    // set) whose child references point into the crown where they can.  crown_root: the root's crown reference.
    const uint32_t* crown;
    uint32_t crown_n, crown_root;
};
constexpr uint32_t NEAR_NONE = 0xFFFFFFFFu;

}  // namespace rtd
