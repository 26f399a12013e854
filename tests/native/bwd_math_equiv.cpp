// Host-side bit-equivalence check of the two formulations in splatter360_amd/csrc/s360_bwd_math.h (built and run by
// tests/test_bwd_math.py with the same -ffp-contract=off as the library).  Exit code 0 = every output and every
// state variable is bit-identical over all trials, including zeros, denormal-range and large inputs.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "s360_bwd_math.h"

static uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

int main(int argc, char** argv) {
    const long trials = argc > 1 ? std::atol(argv[1]) : 2000000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<float> u01(0.f, 1.f), sym(-1.f, 1.f);
    auto pick = [&](float scale) {   // mostly smooth values, sometimes exact zeros / tiny / huge magnitudes
        const int r = (int)(rng() % 16);
        if (r == 0) return 0.0f;
        if (r == 1) return sym(rng) * 1e-30f * scale;
        if (r == 2) return sym(rng) * 1e6f * scale;
        return sym(rng) * scale;
    };
    long bad = 0;
    for (long t = 0; t < trials; ++t) {
        s360::BwdPixel a = {u01(rng), pick(2.f), pick(2.f), pick(2.f), pick(2.f), pick(2.f), pick(2.f), (rng() % 4) ? u01(rng) * 0.99f : 0.f};
        s360::BwdPixel b = a;
        const s360::BwdConst k = {pick(1.f), pick(1.f), pick(1.f), u01(rng), pick(1.f)};
        const bool active = rng() % 5 != 0;
        const float alpha = active ? u01(rng) * 0.99f : 0.f, G = active ? u01(rng) : 0.f;
        const float om = 1.f - alpha, rcp = 1.f / om;
        const float cA = -u01(rng) * 3.f, cB = pick(2.f), cC = -u01(rng) * 3.f, op = u01(rng);
        const float c0 = pick(3.f), c1 = pick(3.f), c2 = pick(3.f), dx = pick(20.f), dy = pick(20.f);
        s360::BwdOut oa, ob;
        s360::bwd_entry_scalar(a, k, alpha, G, rcp, cA, cB, cC, op, c0, c1, c2, dx, dy, oa);
        s360::bwd_entry_packed(b, k, alpha, G, rcp, cA, cB, cC, op, c0, c1, c2, dx, dy, ob);
        const float* fa[] = {&oa.g_x, &oa.g_y, &oa.g_A, &oa.g_B, &oa.g_C, &oa.g_op, &oa.g_r, &oa.g_g, &oa.g_b,
                             &a.T, &a.acc0, &a.acc1, &a.acc2, &a.lc0, &a.lc1, &a.lc2, &a.last_alpha};
        const float* fb[] = {&ob.g_x, &ob.g_y, &ob.g_A, &ob.g_B, &ob.g_C, &ob.g_op, &ob.g_r, &ob.g_g, &ob.g_b,
                             &b.T, &b.acc0, &b.acc1, &b.acc2, &b.lc0, &b.lc1, &b.lc2, &b.last_alpha};
        for (int i = 0; i < 17; ++i)
            if (bits(*fa[i]) != bits(*fb[i])) {
                if (bad < 5) std::printf("mismatch trial %ld field %d: %a vs %a\n", t, i, *fa[i], *fb[i]);
                ++bad;
            }
    }
    std::printf("%ld trials, %ld mismatching values\n", trials, bad);
    return bad ? 1 : 0;
}
