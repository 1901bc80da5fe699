// Host-logic check (no GPU) of what the C++ mirror of the reference's scene API accepts: the reference lets
// TransformedObject<T> and ConstantMedium<T> wrap any Hittable (shapes.rs:26-41, 624-646); the device takes up to four
// transforms stacked inside each other, and media in the world or inside such transforms whose boundary carries at most
// one transform of its own.
// Anything else must be refused with a message that says so -- never flattened into something else.
#include <cstdio>
#include <cstring>
#include <string>

#include "../../raytracer-rs_amd/host/rt_scene.hpp"

using namespace rt;

template <class F>
static bool refuses(const char* what, F build, const char* needle) {
    FlatScene fs;
    try {
        build(fs);
    } catch (const std::exception& e) {
        if (std::strstr(e.what(), needle)) return true;
        std::printf("FAIL %s: message '%s' lacks '%s'\n", what, e.what(), needle);
        return false;
    }
    std::printf("FAIL %s: accepted\n", what);
    return false;
}

int main() {
    const Mat4 shift = Mat4::from_translation(Vec3(1.0f, 2.0f, 3.0f));
    const Mat4 turn = Mat4::from_rotation_y(0.3f);
    bool ok = true;
    {  // one level: accepted, one instance group, ids 0..5 for the box sides
        FlatScene fs;
        auto inst = TransformedObject<Box2>::new_(Box2::new_(1.0f, 2.0f, 3.0f, LambertianMat::from_color(Vec3A(0.5f, 0.5f, 0.5f))), shift);
        inst.flatten(fs, 0);
        if (fs.groups.size() != 1 || fs.rect_id.size() != 6 || fs.rect_group[0] != 1) {
            std::printf("FAIL one level\n");
            ok = false;
        }
    }
    {  // transform inside transform: two instance groups, the inner one names the outer one as its parent
        FlatScene fs;
        auto inner = TransformedObject<Sphere>::new_(Sphere::new_(Vec3A(0, 0, 0), 1.0f, DielectricMat::new_(1.5f)), turn);
        TransformedObject<TransformedObject<Sphere>>::new_(inner, shift).flatten(fs, 0);
        if (fs.groups.size() != 2 || fs.groups[0].parent != 0 || fs.groups[1].parent != 1 || fs.sph_group.size() != 1 || fs.sph_group[0] != 2) {
            std::printf("FAIL two levels\n");
            ok = false;
        }
    }
    {  // four stacked transforms are taken, five are refused
        using T1 = TransformedObject<Sphere>;
        using T2 = TransformedObject<T1>;
        using T3 = TransformedObject<T2>;
        using T4 = TransformedObject<T3>;
        using T5 = TransformedObject<T4>;
        auto t1 = T1::new_(Sphere::new_(Vec3A(0, 0, 0), 1.0f, DielectricMat::new_(1.5f)), turn);
        auto t4 = T4::new_(T3::new_(T2::new_(t1, shift), turn), shift);
        FlatScene fs;
        t4.flatten(fs, 0);
        if (fs.groups.size() != 4 || fs.groups[3].parent != 3) {
            std::printf("FAIL four levels\n");
            ok = false;
        }
        ok &= refuses("five stacked transforms", [&](FlatScene& f5) { T5::new_(t4, turn).flatten(f5, 0); }, "more than 4 deep");
    }
    {  // medium inside transform: an (empty) instance group and a medium that names it as its parent
        FlatScene fs;
        auto fog = ConstantMedium<Sphere>::from_color(Sphere::new_(Vec3A(0, 0, 0), 1.0f, DielectricMat::new_(1.5f)), 0.1f, Vec3A(1, 1, 1));
        TransformedObject<ConstantMedium<Sphere>>::new_(fog, shift).flatten(fs, 0);
        if (fs.groups.size() != 2 || fs.groups[0].kind != RT_GROUP_INSTANCE || fs.groups[1].kind != RT_GROUP_MEDIUM || fs.groups[1].parent != 1 ||
            fs.sph_group.size() != 1 || fs.sph_group[0] != 2) {
            std::printf("FAIL medium inside transform\n");
            ok = false;
        }
    }
    ok &= refuses("medium inside a medium's boundary", [&](FlatScene& fs) {
        auto fog = ConstantMedium<Sphere>::from_color(Sphere::new_(Vec3A(0, 0, 0), 1.0f, DielectricMat::new_(1.5f)), 0.1f, Vec3A(1, 1, 1));
        ConstantMedium<ConstantMedium<Sphere>>::from_color(fog, 0.1f, Vec3A(1, 1, 1)).flatten(fs, 0);
    }, "nested");
    ok &= refuses("transform inside the transform a medium wraps", [&](FlatScene& fs) {
        auto inner = TransformedObject<Sphere>::new_(Sphere::new_(Vec3A(0, 0, 0), 1.0f, DielectricMat::new_(1.5f)), turn);
        auto outer = TransformedObject<TransformedObject<Sphere>>::new_(inner, shift);
        ConstantMedium<TransformedObject<TransformedObject<Sphere>>>::from_color(outer, 0.1f, Vec3A(1, 1, 1)).flatten(fs, 0);
    }, "nested");
    std::printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
