// score_generate.hpp -- synthetic multi-robot Manhattan-world RA-SLAM graphs made where they are solved (the
// "device-resident batched generator").  A Monte-Carlo study over such worlds (BASELINE configs[4]) otherwise spends its time
// in the generator: score_amd.manhattan.make_manhattan builds 4 x 1000 poses in ~30 ms of Python, the solver needs 0.5 ms.
//
// Statistics (measured from the reference's shipped fixture examples/manhattan/factor_graph.pickle; the same as
// score_amd/manhattan.py): integer-lattice walks with unit steps on a grid of side `side`, headings in {0, +-pi/2, pi}, the next
// heading drawn from {straight 0.81, left 0.09, right 0.09, back 0.01} in weighted random order until the following step stays
// inside the grid; robot 0 starts at the origin with identity heading (the pinned pose), the others at a random lattice point
// and heading whose first step stays inside; odometry = (1, 0, turn) in the base frame with noise sigma_t on both translation
// components and sigma_theta on the angle (precisions 1 / sigma^2); beacons at random lattice points; at every timestep every
// robot-beacon pair and every robot-robot pair is measured with probability p_range, distance = max(0, true + sigma_range N(0,1)),
// precision 1 / sigma_range^2; no loop closures, no priors.  Order of the measurements = make_manhattan's: odometry chain by
// chain; ranges robot by robot (time-major, beacon-minor), then the robot pairs (a < b) in time order.
//
// Randomness: Philox4x32-10 (counter based), key = seed + trial, counter = (purpose, robot / group, step, index): trial t of any
// call is the same world whatever the batch it is generated in, and every value can be recomputed independently -- the count
// and fill passes of the ranges draw the same numbers twice.  One set of functions serves the HIP kernels and the host loops of
// the CPU twin (the specification the kernels are tested against: integers bit-equal, reals to rounding of the math library).
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../../include/score_hip.h"

#if defined(__HIPCC__)
#define SCORE_GEN_HD __host__ __device__
#else
#define SCORE_GEN_HD
#endif

namespace score {

struct GenSpec {  // (score_manhattan_spec of the ABI)
    int32_t n_robots, n_poses, n_beacons, side;
    double p_range, sigma_t, sigma_theta, sigma_range;
    uint64_t seed;
    int32_t dim = 2;  // 2: the shipped fixture's worlds; 3: their counterpart on the lattice of a cube (round 6, below)
};

struct Philox4 { uint32_t v[4]; };
SCORE_GEN_HD inline Philox4 philox4x32_10(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1, y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = x0; o.v[1] = x1; o.v[2] = x2; o.v[3] = x3;
    return o;
}
enum GenPurpose : uint32_t { GEN_START = 1, GEN_TURN = 2, GEN_ODOM_A = 3, GEN_ODOM_B = 4, GEN_BEACON = 5, GEN_HIT_RB = 6, GEN_NOISE_RB = 7, GEN_HIT_RR = 8, GEN_NOISE_RR = 9,
                            GEN_TURN_B = 10, GEN_ODOM_C = 11, GEN_ODOM_D = 12 };  // (3-D: two more turn draws, the rotation noise)

SCORE_GEN_HD inline double gen_u01(uint32_t hi, uint32_t lo) {  // [0, 1): 53 bits
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}
SCORE_GEN_HD inline double gen_u32(uint32_t a) { return (double)a * (1.0 / 4294967296.0); }  // [0, 1): 32 bits (categorical draws)
// two standard normals from one Philox block (Box-Muller)
SCORE_GEN_HD inline void gen_normal2(const Philox4& p, double& n0, double& n1) {
    const double u = 1.0 - gen_u01(p.v[0], p.v[1]);  // (0, 1]
    const double v = gen_u01(p.v[2], p.v[3]);
    const double r = std::sqrt(-2.0 * std::log(u)), a = 6.283185307179586476925286766559 * v;
    n0 = r * std::cos(a);
    n1 = r * std::sin(a);
}
SCORE_GEN_HD inline uint32_t gen_below(uint32_t x, uint32_t n) { return (uint32_t)(((uint64_t)x * n) >> 32); }  // [0, n)

SCORE_GEN_HD inline void gen_dir(int h, int& dx, int& dy) {
    dx = (h == 0) - (h == 2);
    dy = (h == 1) - (h == 3);
}
SCORE_GEN_HD inline bool gen_inside(int x, int y, int side) { return x >= 0 && x <= side && y >= 0 && y <= side; }

// start of robot r (r = 0: the origin, heading 0)
SCORE_GEN_HD inline void gen_start(const GenSpec& S, uint64_t key, int r, int& x, int& y, int& h) {
    if (r == 0) { x = 0; y = 0; h = 0; return; }
    for (uint32_t attempt = 0; attempt < 256; ++attempt) {
        const Philox4 p = philox4x32_10(key, GEN_START, (uint32_t)r, attempt, 0);
        x = (int)gen_below(p.v[0], (uint32_t)S.side + 1); y = (int)gen_below(p.v[1], (uint32_t)S.side + 1); h = (int)(p.v[2] >> 30);
        int dx, dy;
        gen_dir(h, dx, dy);
        if (gen_inside(x + dx, y + dy, S.side)) return;
    }
    x = 0; y = 0; h = 0;  // (a grid of side >= 1 always has a feasible draw long before)
}
// heading after arriving at (x, y) with heading h_prev at step i: weighted random order of {straight, left, right, back}
// (scalars and selects only: arrays indexed by the draw would live in scratch memory on the device -- the walk is the one
//  sequential loop of the generator)
SCORE_GEN_HD inline int gen_next_heading(const GenSpec& S, uint64_t key, int r, int i, int x, int y, int h_prev) {
    const Philox4 p = philox4x32_10(key, GEN_TURN, (uint32_t)r, (uint32_t)i, 0);
    double w0 = 0.81, w1 = 0.09, w2 = 0.09, w3 = 0.01;  // straight, left (+1), right (-1 = +3), back (+2)
    double tot = 1.0;
#pragma unroll
    for (int draw = 0; draw < 4; ++draw) {
        const uint32_t bits = draw == 0 ? p.v[0] : draw == 1 ? p.v[1] : draw == 2 ? p.v[2] : p.v[3];
        const double u = gen_u32(bits) * tot;
        // the first live weight whose running sum exceeds u; the last live one otherwise
        int pick = -1;
        double acc = 0.0;
        bool found = false;
        if (w0 > 0.0) { pick = 0; acc += w0; found = u < acc; }
        if (!found && w1 > 0.0) { pick = 1; acc += w1; found = u < acc; }
        if (!found && w2 > 0.0) { pick = 2; acc += w2; found = u < acc; }
        if (!found && w3 > 0.0) { pick = 3; acc += w3; found = u < acc; }
        const int turn = pick == 0 ? 0 : pick == 1 ? 1 : pick == 2 ? 3 : 2;
        const int h = (h_prev + turn) & 3;
        int dx, dy;
        gen_dir(h, dx, dy);
        if (gen_inside(x + dx, y + dy, S.side)) return h;
        const double wp = pick == 0 ? w0 : pick == 1 ? w1 : pick == 2 ? w2 : w3;
        tot -= wp;
        if (pick == 0) w0 = 0.0; else if (pick == 1) w1 = 0.0; else if (pick == 2) w2 = 0.0; else w3 = 0.0;
    }
    return (h_prev + 2) & 3;
}
SCORE_GEN_HD inline double gen_wrap(double th) { return std::atan2(std::sin(th), std::cos(th)); }

// The walk of robot r of the world `key`: true poses (x, y, heading index).  Sequential in the steps (the next heading depends on
// where the robot stands); everything else about a world is independent per item.
SCORE_GEN_HD inline void gen_walk(const GenSpec& S, uint64_t key, int r, int32_t* px, int32_t* py, int32_t* ph) {
    int x, y, h;
    gen_start(S, key, r, x, y, h);
    px[0] = x; py[0] = y; ph[0] = h;
    for (int i = 1; i < S.n_poses; ++i) {
        int dx, dy;
        gen_dir(h, dx, dy);
        x += dx; y += dy;
        h = gen_next_heading(S, key, r, i, x, y, h);
        px[i] = x; py[i] = y; ph[i] = h;
    }
}
// Odometry edge e of robot r (pose e -> e + 1): a unit step forward in the base frame, then the turn, with noise.
// rel_* point at the edge's slots; pose0 = index of the robot's first pose; h, hn: headings at the two poses.
SCORE_GEN_HD inline void gen_odom(const GenSpec& S, uint64_t key, int r, int e, int32_t pose0, int h, int hn, int32_t* rel_base, int32_t* rel_to,
                                  double* rel_t, double* rel_R, double* rel_kappa, double* rel_tau) {
    const double dth = (double)(((hn - h + 1) & 3) - 1) * 1.5707963267948966192313216916398;
    double n0, n1, n2, n3;
    gen_normal2(philox4x32_10(key, GEN_ODOM_A, (uint32_t)r, (uint32_t)e, 0), n0, n1);
    gen_normal2(philox4x32_10(key, GEN_ODOM_B, (uint32_t)r, (uint32_t)e, 0), n2, n3);
    const double th = gen_wrap(dth + S.sigma_theta * n2);
    const double c = std::cos(th), s = std::sin(th);
    rel_base[0] = pose0 + e; rel_to[0] = pose0 + e + 1;
    rel_t[0] = 1.0 + S.sigma_t * n0; rel_t[1] = S.sigma_t * n1;
    rel_R[0] = c; rel_R[1] = -s; rel_R[2] = s; rel_R[3] = c;
    rel_kappa[0] = 1.0 / (S.sigma_t * S.sigma_t); rel_tau[0] = 1.0 / (S.sigma_theta * S.sigma_theta);
}
// ---------------------------------------------------------------------------------------------------------------------------
// 3-D worlds (round 6; the reference's model is dimension-generic, gurobi_utils.py:37-50, and ships no 3-D data: the statistics
// are the 2-D fixture's carried over).  A robot walks the integer lattice of the cube [0, side]^3 with an axis-aligned
// orientation: one unit step along its body x axis per pose, then a turn drawn from {straight 0.80, yaw left 0.045, yaw right
// 0.045, pitch up 0.045, pitch down 0.045, back 0.02} in weighted random order until the following step stays inside the cube.
// An orientation is the pair (forward axis f, up axis u), axes coded 0..5 = +x -x +y -y +z -z, as o = 6 f + u; the body frame is
// (forward, left = up x forward, up), R = [forward | left | up] maps body to world.  Odometry = (1, 0, 0) + sigma_t N(0, I) in the
// base frame, rotation R_i' R_{i+1} (exact integers) times Exp(sigma_theta N(0, I)) (Rodrigues); ranges as in 2-D with Euclidean
// distances in space.  Robot 0 starts at the origin with the identity orientation (the pinned pose).
// ---------------------------------------------------------------------------------------------------------------------------
SCORE_GEN_HD inline void gen_axis(int code, int (&v)[3]) {
    v[0] = v[1] = v[2] = 0;
    v[code >> 1] = (code & 1) ? -1 : 1;
}
SCORE_GEN_HD inline int gen_axis_code(const int (&v)[3]) {
    return v[0] ? (v[0] > 0 ? 0 : 1) : v[1] ? (v[1] > 0 ? 2 : 3) : (v[2] > 0 ? 4 : 5);
}
SCORE_GEN_HD inline int gen_cross_code(int a, int b) {  // axis code of a x b (a, b perpendicular)
    int x[3], y[3];
    gen_axis(a, x); gen_axis(b, y);
    const int c[3] = {x[1] * y[2] - x[2] * y[1], x[2] * y[0] - x[0] * y[2], x[0] * y[1] - x[1] * y[0]};
    return gen_axis_code(c);
}
// R (row-major 3 x 3 integers) of orientation o: columns forward, left, up
SCORE_GEN_HD inline void gen_rot3(int o, int (&R)[9]) {
    const int f = o / 6, u = o - 6 * f, l = gen_cross_code(u, f);
    int cf[3], cl[3], cu[3];
    gen_axis(f, cf); gen_axis(l, cl); gen_axis(u, cu);
    for (int k = 0; k < 3; ++k) { R[3 * k] = cf[k]; R[3 * k + 1] = cl[k]; R[3 * k + 2] = cu[k]; }
}
// orientation after turn t (0 straight, 1 yaw left, 2 yaw right, 3 pitch up, 4 pitch down, 5 back)
SCORE_GEN_HD inline int gen_turned3(int o, int t) {
    const int f = o / 6, u = o - 6 * f, l = gen_cross_code(u, f);
    const int nf = t == 0 ? f : t == 1 ? l : t == 2 ? (l ^ 1) : t == 3 ? u : t == 4 ? (u ^ 1) : (f ^ 1);
    const int nu = t == 3 ? (f ^ 1) : t == 4 ? f : u;
    return 6 * nf + nu;
}
SCORE_GEN_HD inline bool gen_inside3(int x, int y, int z, int side) { return x >= 0 && x <= side && y >= 0 && y <= side && z >= 0 && z <= side; }
SCORE_GEN_HD inline bool gen_step_inside3(int x, int y, int z, int o, int side) {
    int d[3];
    gen_axis(o / 6, d);
    return gen_inside3(x + d[0], y + d[1], z + d[2], side);
}
SCORE_GEN_HD inline void gen_start3(const GenSpec& S, uint64_t key, int r, int& x, int& y, int& z, int& o) {
    if (r == 0) { x = 0; y = 0; z = 0; o = 6 * 0 + 4; return; }  // forward +x, up +z: the identity
    for (uint32_t attempt = 0; attempt < 256; ++attempt) {
        const Philox4 p = philox4x32_10(key, GEN_START, (uint32_t)r, attempt, 0);
        x = (int)gen_below(p.v[0], (uint32_t)S.side + 1); y = (int)gen_below(p.v[1], (uint32_t)S.side + 1); z = (int)gen_below(p.v[2], (uint32_t)S.side + 1);
        const int f = (int)gen_below(p.v[3] >> 16 << 16, 6), k = (int)gen_below(p.v[3] << 16, 4);  // one of the 24 orientations
        // the k-th axis perpendicular to f, in code order
        int u = -1, seen = 0;
        for (int c = 0; c < 6; ++c) {
            if ((c >> 1) == (f >> 1)) continue;
            if (seen == k) { u = c; break; }
            ++seen;
        }
        o = 6 * f + u;
        if (gen_step_inside3(x, y, z, o, S.side)) return;
    }
    x = 0; y = 0; z = 0; o = 4;
}
SCORE_GEN_HD inline int gen_next_orientation3(const GenSpec& S, uint64_t key, int r, int i, int x, int y, int z, int o_prev) {
    const Philox4 pa = philox4x32_10(key, GEN_TURN, (uint32_t)r, (uint32_t)i, 0), pb = philox4x32_10(key, GEN_TURN_B, (uint32_t)r, (uint32_t)i, 0);
    double w0 = 0.80, w1 = 0.045, w2 = 0.045, w3 = 0.045, w4 = 0.045, w5 = 0.02;
    double tot = 1.0;
#pragma unroll
    for (int draw = 0; draw < 6; ++draw) {
        const uint32_t bits = draw == 0 ? pa.v[0] : draw == 1 ? pa.v[1] : draw == 2 ? pa.v[2] : draw == 3 ? pa.v[3] : draw == 4 ? pb.v[0] : pb.v[1];
        const double u = gen_u32(bits) * tot;
        int pick = -1;
        double acc = 0.0;
        bool found = false;
        if (w0 > 0.0) { pick = 0; acc += w0; found = u < acc; }
        if (!found && w1 > 0.0) { pick = 1; acc += w1; found = u < acc; }
        if (!found && w2 > 0.0) { pick = 2; acc += w2; found = u < acc; }
        if (!found && w3 > 0.0) { pick = 3; acc += w3; found = u < acc; }
        if (!found && w4 > 0.0) { pick = 4; acc += w4; found = u < acc; }
        if (!found && w5 > 0.0) { pick = 5; acc += w5; found = u < acc; }
        const int o = gen_turned3(o_prev, pick);
        if (gen_step_inside3(x, y, z, o, S.side)) return o;
        const double wp = pick == 0 ? w0 : pick == 1 ? w1 : pick == 2 ? w2 : pick == 3 ? w3 : pick == 4 ? w4 : w5;
        tot -= wp;
        if (pick == 0) w0 = 0.0; else if (pick == 1) w1 = 0.0; else if (pick == 2) w2 = 0.0; else if (pick == 3) w3 = 0.0; else if (pick == 4) w4 = 0.0; else w5 = 0.0;
    }
    return gen_turned3(o_prev, 5);
}
SCORE_GEN_HD inline void gen_walk3(const GenSpec& S, uint64_t key, int r, int32_t* px, int32_t* py, int32_t* pz, int32_t* po) {
    int x, y, z, o;
    gen_start3(S, key, r, x, y, z, o);
    px[0] = x; py[0] = y; pz[0] = z; po[0] = o;
    for (int i = 1; i < S.n_poses; ++i) {
        int d[3];
        gen_axis(o / 6, d);
        x += d[0]; y += d[1]; z += d[2];
        o = gen_next_orientation3(S, key, r, i, x, y, z, o);
        px[i] = x; py[i] = y; pz[i] = z; po[i] = o;
    }
}
// Exp of a rotation vector (Rodrigues), row-major
SCORE_GEN_HD inline void gen_exp3(double wx, double wy, double wz, double (&E)[9]) {
    const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
    double a, b;  // E = I + a [w]x + b [w]x^2
    if (th < 1e-8) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; }
    else { a = std::sin(th) / th; b = (1.0 - std::cos(th)) / th2; }
    E[0] = 1.0 - b * (wy * wy + wz * wz); E[1] = -a * wz + b * wx * wy;       E[2] = a * wy + b * wx * wz;
    E[3] = a * wz + b * wx * wy;          E[4] = 1.0 - b * (wx * wx + wz * wz); E[5] = -a * wx + b * wy * wz;
    E[6] = -a * wy + b * wx * wz;         E[7] = a * wx + b * wy * wz;        E[8] = 1.0 - b * (wx * wx + wy * wy);
}
SCORE_GEN_HD inline void gen_odom3(const GenSpec& S, uint64_t key, int r, int e, int32_t pose0, int o, int on, int32_t* rel_base, int32_t* rel_to,
                                   double* rel_t, double* rel_R, double* rel_kappa, double* rel_tau) {
    int Ri[9], Rn[9];
    gen_rot3(o, Ri); gen_rot3(on, Rn);
    double n0, n1, n2, n3, n4, n5, n6, n7;
    gen_normal2(philox4x32_10(key, GEN_ODOM_A, (uint32_t)r, (uint32_t)e, 0), n0, n1);
    gen_normal2(philox4x32_10(key, GEN_ODOM_B, (uint32_t)r, (uint32_t)e, 0), n2, n3);
    gen_normal2(philox4x32_10(key, GEN_ODOM_C, (uint32_t)r, (uint32_t)e, 0), n4, n5);
    gen_normal2(philox4x32_10(key, GEN_ODOM_D, (uint32_t)r, (uint32_t)e, 0), n6, n7);
    (void)n3; (void)n7;
    double E[9];
    gen_exp3(S.sigma_theta * n4, S.sigma_theta * n5, S.sigma_theta * n6, E);
    rel_base[0] = pose0 + e; rel_to[0] = pose0 + e + 1;
    rel_t[0] = 1.0 + S.sigma_t * n0; rel_t[1] = S.sigma_t * n1; rel_t[2] = S.sigma_t * n2;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) {
                int turn = 0;  // (R_i' R_n)[a][k]
                for (int m = 0; m < 3; ++m) turn += Ri[3 * m + a] * Rn[3 * m + k];
                acc += (double)turn * E[3 * k + b];
            }
            rel_R[3 * a + b] = acc;
        }
    rel_kappa[0] = 1.0 / (S.sigma_t * S.sigma_t); rel_tau[0] = 1.0 / (S.sigma_theta * S.sigma_theta);
}
SCORE_GEN_HD inline void gen_beacon3(const GenSpec& S, uint64_t key, int b, int32_t& x, int32_t& y, int32_t& z) {
    const Philox4 p = philox4x32_10(key, GEN_BEACON, (uint32_t)b, 0, 0);
    x = (int32_t)gen_below(p.v[0], (uint32_t)S.side + 1);
    y = (int32_t)gen_below(p.v[1], (uint32_t)S.side + 1);
    z = (int32_t)gen_below(p.v[2], (uint32_t)S.side + 1);
}

SCORE_GEN_HD inline void gen_beacon(const GenSpec& S, uint64_t key, int b, int32_t& x, int32_t& y) {
    const Philox4 p = philox4x32_10(key, GEN_BEACON, (uint32_t)b, 0, 0);
    x = (int32_t)gen_below(p.v[0], (uint32_t)S.side + 1);
    y = (int32_t)gen_below(p.v[1], (uint32_t)S.side + 1);
}
// groups of range measurements: g < n_robots: robot g against the beacons; then the robot pairs (a < b) in lexicographic order
SCORE_GEN_HD inline int gen_groups(const GenSpec& S) { return S.n_robots + S.n_robots * (S.n_robots - 1) / 2; }
SCORE_GEN_HD inline void gen_pair(const GenSpec& S, int g, int& a, int& b) {
    int k = g - S.n_robots;
    for (a = 0; a < S.n_robots; ++a) {
        const int row = S.n_robots - 1 - a;
        if (k < row) { b = a + 1 + k; return; }
        k -= row;
    }
    a = 0; b = 1;
}
SCORE_GEN_HD inline bool gen_hit_rb(const GenSpec& S, uint64_t key, int r, int t, int b) {
    const Philox4 p = philox4x32_10(key, GEN_HIT_RB, (uint32_t)r, (uint32_t)t, (uint32_t)(b >> 2));
    return gen_u32(p.v[b & 3]) < S.p_range;
}
SCORE_GEN_HD inline bool gen_hit_rr(const GenSpec& S, uint64_t key, int g, int t) {
    const Philox4 p = philox4x32_10(key, GEN_HIT_RR, (uint32_t)g, (uint32_t)t, 0);
    return gen_u32(p.v[0]) < S.p_range;
}
// measurements of group g at time t: returns the count; with out pointers (positioned at this (g, t)'s first slot) writes them
// (pz / bz: the third coordinate of a 3-D world, null in 2-D)
SCORE_GEN_HD inline int gen_ranges_at(const GenSpec& S, uint64_t key, int g, int t, const int32_t* px, const int32_t* py, const int32_t* bx,
                                  const int32_t* by, int32_t* ra, int32_t* rb, double* dist, double* prec, const int32_t* pz = nullptr,
                                  const int32_t* bz = nullptr) {
    const int T = S.n_poses, Np = S.n_robots * T;
    const double w = 1.0 / (S.sigma_range * S.sigma_range);
    int n = 0;
    if (g < S.n_robots) {
        for (int b = 0; b < S.n_beacons; ++b) {
            if (!gen_hit_rb(S, key, g, t, b)) continue;
            if (ra) {
                const double ddx = (double)(px[g * T + t] - bx[b]), ddy = (double)(py[g * T + t] - by[b]);
                const double ddz = pz ? (double)(pz[g * T + t] - bz[b]) : 0.0;
                double n0, n1;
                gen_normal2(philox4x32_10(key, GEN_NOISE_RB, (uint32_t)g, (uint32_t)t, (uint32_t)b), n0, n1);
                const double m = std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz) + S.sigma_range * n0;
                ra[n] = g * T + t; rb[n] = Np + b; dist[n] = m > 0.0 ? m : 0.0; prec[n] = w;
            }
            ++n;
        }
    } else if (gen_hit_rr(S, key, g, t)) {
        if (ra) {
            int a, b;
            gen_pair(S, g, a, b);
            const double ddx = (double)(px[a * T + t] - px[b * T + t]), ddy = (double)(py[a * T + t] - py[b * T + t]);
            const double ddz = pz ? (double)(pz[a * T + t] - pz[b * T + t]) : 0.0;
            double n0, n1;
            gen_normal2(philox4x32_10(key, GEN_NOISE_RR, (uint32_t)g, (uint32_t)t, 0), n0, n1);
            const double m = std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz) + S.sigma_range * n0;
            ra[0] = a * T + t; rb[0] = b * T + t; dist[0] = m > 0.0 ? m : 0.0; prec[0] = w;
        }
        n = 1;
    }
    return n;
}

inline void gen_check_spec(const GenSpec& S, int count) {
    if (count <= 0) throw std::runtime_error("score_generate_manhattan: count must be positive");
    if (S.n_robots < 1 || S.n_robots > 64 || S.n_poses < 2 || S.n_beacons < 0 || S.n_beacons > 4096 || S.side < 1)
        throw std::runtime_error("score_generate_manhattan: need 1..64 robots, >= 2 poses, 0..4096 beacons, side >= 1");
    if (S.dim != 2 && S.dim != 3) throw std::runtime_error("score_generate_manhattan: dim must be 2 or 3");
    if (!(S.p_range >= 0.0 && S.p_range <= 1.0) || !(S.sigma_t > 0.0) || !(S.sigma_theta > 0.0) || !(S.sigma_range > 0.0))
        throw std::runtime_error("score_generate_manhattan: p_range in [0, 1], positive noise levels");
    const int64_t Np = (int64_t)S.n_robots * S.n_poses;
    if (Np * count >= ((int64_t)1 << 30) || (int64_t)gen_groups(S) * S.n_poses * count >= ((int64_t)1 << 30))
        throw std::runtime_error("score_generate_manhattan: batch too large");
    // the ranges: a robot-beacon slot emits up to n_beacons measurements, a timestep's robot pairs R (R - 1) / 2; offsets and
    // the device scan are 32-bit
    const int64_t worst = (int64_t)count * S.n_poses * ((int64_t)S.n_robots * S.n_beacons + (int64_t)S.n_robots * (S.n_robots - 1) / 2);
    if (worst >= ((int64_t)1 << 30))
        throw std::runtime_error("score_generate_manhattan: batch too large (more than 2^30 possible range measurements): draw fewer worlds per call");
}

// (storage that is NOT zero-filled when sized: every element is written by the generator; a std::vector's resize touches
//  22 MB for 64 worlds of 4 x 1000 poses before the first useful byte arrives)
template <class T>
struct RawVec {
    T* p = nullptr;
    size_t n = 0;
    RawVec() = default;
    RawVec(const RawVec&) = delete;
    RawVec& operator=(const RawVec&) = delete;
    RawVec(RawVec&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    RawVec& operator=(RawVec&& o) noexcept { if (this != &o) { delete[] p; p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~RawVec() { delete[] p; }
    void resize(size_t count) { delete[] p; p = count ? new T[count] : nullptr; n = count; }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

// A generated batch on the host: the flat arrays of `count` score_graph structs (trial after trial) + the ground truth.
struct GeneratedBatch {
    GenSpec S{};
    int32_t count = 0;
    std::vector<int32_t> chain_len;                       // n_robots entries (shared by every trial)
    RawVec<int32_t> px, py, ph, bx, by;                   // truth: lattice positions, heading indices (3-D: orientation codes), beacons
    RawVec<int32_t> pz, bz;                               // (3-D worlds only)
    RawVec<int32_t> rel_base, rel_to, ra, rb;
    RawVec<double> rel_t, rel_R, rel_kappa, rel_tau, dist, prec;
    std::vector<int64_t> rng_first;                       // count + 1: a trial's ranges in ra / rb / dist / prec
    int64_t edges() const { return (int64_t)S.n_robots * (S.n_poses - 1); }
    void size_fixed() {
        const size_t R = (size_t)S.n_robots, T = (size_t)S.n_poses, E = (size_t)edges(), c = (size_t)count;
        chain_len.assign(R, (int32_t)T);
        const size_t d = (size_t)S.dim;
        px.resize(c * R * T); py.resize(c * R * T); ph.resize(c * R * T);
        bx.resize(c * (size_t)S.n_beacons); by.resize(c * (size_t)S.n_beacons);
        if (d == 3) { pz.resize(c * R * T); bz.resize(c * (size_t)S.n_beacons); }
        rel_base.resize(c * E); rel_to.resize(c * E); rel_t.resize(d * c * E); rel_R.resize(d * d * c * E); rel_kappa.resize(c * E); rel_tau.resize(c * E);
        rng_first.assign(c + 1, 0);
    }
    void size_ranges(int64_t total) { ra.resize((size_t)total); rb.resize((size_t)total); dist.resize((size_t)total); prec.resize((size_t)total); }
    void view(int32_t i, score_graph* g) const {
        if (i < 0 || i >= count) throw std::runtime_error("score_generated_graph: index out of range");
        const size_t E = (size_t)edges(), o = (size_t)i * E, r0 = (size_t)rng_first[(size_t)i];
        *g = score_graph{};
        const size_t d = (size_t)S.dim;
        g->dim = S.dim; g->relaxation = 0; g->n_chains = S.n_robots; g->chain_len = chain_len.data(); g->n_landmarks = S.n_beacons;
        g->n_rel = (int64_t)E; g->rel_base = rel_base.data() + o; g->rel_to = rel_to.data() + o; g->rel_t = rel_t.data() + d * o;
        g->rel_R = rel_R.data() + d * d * o; g->rel_kappa = rel_kappa.data() + o; g->rel_tau = rel_tau.data() + o;
        g->n_rng = rng_first[(size_t)i + 1] - rng_first[(size_t)i];
        g->rng_a = ra.data() + r0; g->rng_b = rb.data() + r0; g->rng_dist = dist.data() + r0; g->rng_prec = prec.data() + r0;
        g->n_lprior = 0; g->lprior_lm = nullptr; g->lprior_t = nullptr; g->lprior_prec = nullptr;
    }
    // 2-D: poses n_robots * n_poses x (x, y, theta), beacons n_beacons x (x, y);
    // 3-D: poses x (x, y, z, R row-major: 12 values), beacons x (x, y, z)
    void truth(int32_t i, double* poses, double* beacons) const {
        if (i < 0 || i >= count) throw std::runtime_error("score_generated_truth: index out of range");
        const size_t Np = (size_t)S.n_robots * S.n_poses, o = (size_t)i * Np, bo = (size_t)i * (size_t)S.n_beacons;
        if (S.dim == 3) {
            if (poses)
                for (size_t k = 0; k < Np; ++k) {
                    int R[9];
                    gen_rot3(ph[o + k], R);
                    poses[12 * k] = (double)px[o + k]; poses[12 * k + 1] = (double)py[o + k]; poses[12 * k + 2] = (double)pz[o + k];
                    for (int e = 0; e < 9; ++e) poses[12 * k + 3 + e] = (double)R[e];
                }
            if (beacons)
                for (size_t b = 0; b < (size_t)S.n_beacons; ++b) {
                    beacons[3 * b] = (double)bx[bo + b]; beacons[3 * b + 1] = (double)by[bo + b]; beacons[3 * b + 2] = (double)bz[bo + b];
                }
            return;
        }
        if (poses)
            for (size_t k = 0; k < Np; ++k) {
                poses[3 * k] = (double)px[o + k]; poses[3 * k + 1] = (double)py[o + k];
                poses[3 * k + 2] = gen_wrap((double)ph[o + k] * 1.5707963267948966192313216916398);
            }
        if (beacons)
            for (size_t b = 0; b < (size_t)S.n_beacons; ++b) { beacons[2 * b] = (double)bx[bo + b]; beacons[2 * b + 1] = (double)by[bo + b]; }
    }
};

// the generator as host loops (the CPU twin's; the specification of the kernels below)
inline void generate_manhattan_host(const GenSpec& S, int count, GeneratedBatch& B) {
    gen_check_spec(S, count);
    B = GeneratedBatch();
    B.S = S; B.count = count;
    B.size_fixed();
    const int R = S.n_robots, T = S.n_poses, G = gen_groups(S);
    const size_t E = (size_t)B.edges();
    for (int trial = 0; trial < count; ++trial) {
        const uint64_t key = S.seed + (uint64_t)trial;
        for (int r = 0; r < R; ++r) {
            const size_t po = ((size_t)trial * R + r) * T, eo = (size_t)trial * E + (size_t)r * (T - 1);
            if (S.dim == 3) {
                gen_walk3(S, key, r, &B.px[po], &B.py[po], &B.pz[po], &B.ph[po]);
                for (int e = 0; e + 1 < T; ++e)
                    gen_odom3(S, key, r, e, r * T, B.ph[po + e], B.ph[po + e + 1], &B.rel_base[eo + e], &B.rel_to[eo + e], &B.rel_t[3 * (eo + e)],
                              &B.rel_R[9 * (eo + e)], &B.rel_kappa[eo + e], &B.rel_tau[eo + e]);
                continue;
            }
            gen_walk(S, key, r, &B.px[po], &B.py[po], &B.ph[po]);
            for (int e = 0; e + 1 < T; ++e)
                gen_odom(S, key, r, e, r * T, B.ph[po + e], B.ph[po + e + 1], &B.rel_base[eo + e], &B.rel_to[eo + e], &B.rel_t[2 * (eo + e)],
                         &B.rel_R[4 * (eo + e)], &B.rel_kappa[eo + e], &B.rel_tau[eo + e]);
        }
        for (int b = 0; b < S.n_beacons; ++b) {
            const size_t bo = (size_t)trial * S.n_beacons + b;
            if (S.dim == 3) gen_beacon3(S, key, b, B.bx[bo], B.by[bo], B.bz[bo]);
            else gen_beacon(S, key, b, B.bx[bo], B.by[bo]);
        }
    }
    std::vector<int32_t> cnt((size_t)count * G * T);
    int64_t total = 0;
    for (int trial = 0; trial < count; ++trial) {
        B.rng_first[(size_t)trial] = total;
        const uint64_t key = S.seed + (uint64_t)trial;
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) {
                const int n = gen_ranges_at(S, key, g, t, &B.px[(size_t)trial * R * T], &B.py[(size_t)trial * R * T], B.bx.data() + (size_t)trial * S.n_beacons,
                                            B.by.data() + (size_t)trial * S.n_beacons, nullptr, nullptr, nullptr, nullptr);  // (counting: no coordinates read)
                cnt[((size_t)trial * G + g) * T + t] = n;
                total += n;
            }
    }
    B.rng_first[(size_t)count] = total;
    B.size_ranges(total);
    int64_t o = 0;
    for (int trial = 0; trial < count; ++trial) {
        const uint64_t key = S.seed + (uint64_t)trial;
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) {
                const int n = cnt[((size_t)trial * G + g) * T + t];
                if (!n) continue;
                gen_ranges_at(S, key, g, t, &B.px[(size_t)trial * R * T], &B.py[(size_t)trial * R * T], B.bx.data() + (size_t)trial * S.n_beacons,
                              B.by.data() + (size_t)trial * S.n_beacons, &B.ra[(size_t)o], &B.rb[(size_t)o], &B.dist[(size_t)o], &B.prec[(size_t)o],
                              S.dim == 3 ? &B.pz[(size_t)trial * R * T] : nullptr, S.dim == 3 ? B.bz.data() + (size_t)trial * S.n_beacons : nullptr);
                o += n;
            }
    }
}

#if defined(__HIPCC__)
struct GenArgs {
    GenSpec S;
    int32_t count;
    // per trial, robot, pose
    int32_t* px; int32_t* py; int32_t* ph; int32_t* pz; int32_t* bz;  // (pz, bz: 3-D worlds)
    // per trial: n_robots * (n_poses - 1) edges
    int32_t* rel_base; int32_t* rel_to; double* rel_t; double* rel_R; double* rel_kappa; double* rel_tau;
    int32_t* bx; int32_t* by;              // per trial, beacon
    int32_t* cnt;                          // per trial, group, time: measurements there
    const int32_t* off;                    // exclusive scan of cnt
    int32_t* ra; int32_t* rb; double* dist; double* prec;  // concatenated over the batch in scan order
    const int32_t* trial_first;            // per trial: its first range (fill pass: endpoints are trial-local)
};
__global__ __launch_bounds__(64) void k_gen_walk(GenArgs a) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int R = a.S.n_robots, T = a.S.n_poses;
    if (i < a.count * R) {
        const int trial = i / R, r = i - trial * R;
        const size_t po = ((size_t)trial * R + r) * T;
        if (a.S.dim == 3) gen_walk3(a.S, a.S.seed + (uint64_t)trial, r, a.px + po, a.py + po, a.pz + po, a.ph + po);
        else gen_walk(a.S, a.S.seed + (uint64_t)trial, r, a.px + po, a.py + po, a.ph + po);
    }
    const int nb = a.count * a.S.n_beacons;
    if (i < nb) {
        const int trial = i / a.S.n_beacons, b = i - trial * a.S.n_beacons;
        if (a.S.dim == 3) gen_beacon3(a.S, a.S.seed + (uint64_t)trial, b, a.bx[i], a.by[i], a.bz[i]);
        else gen_beacon(a.S, a.S.seed + (uint64_t)trial, b, a.bx[i], a.by[i]);
    }
}
// one thread per odometry edge (after the walks)
__global__ __launch_bounds__(256) void k_gen_odom(GenArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int R = a.S.n_robots, T = a.S.n_poses;
    const int64_t E = (int64_t)R * (T - 1);
    if (i >= a.count * E) return;
    const int trial = (int)(i / E);
    const int rem = (int)(i - trial * E);
    const int r = rem / (T - 1), e = rem - r * (T - 1);
    const size_t po = ((size_t)trial * R + r) * T;
    if (a.S.dim == 3)
        gen_odom3(a.S, a.S.seed + (uint64_t)trial, r, e, r * T, a.ph[po + e], a.ph[po + e + 1], a.rel_base + i, a.rel_to + i, a.rel_t + 3 * i, a.rel_R + 9 * i,
                  a.rel_kappa + i, a.rel_tau + i);
    else
    gen_odom(a.S, a.S.seed + (uint64_t)trial, r, e, r * T, a.ph[po + e], a.ph[po + e + 1], a.rel_base + i, a.rel_to + i, a.rel_t + 2 * i, a.rel_R + 4 * i,
             a.rel_kappa + i, a.rel_tau + i);
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_gen_ranges(GenArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int G = gen_groups(a.S), T = a.S.n_poses, R = a.S.n_robots;
    if (i >= (int64_t)a.count * G * T) return;
    const int trial = (int)(i / ((int64_t)G * T));
    const int rem = (int)(i - (int64_t)trial * G * T);
    const int g = rem / T, t = rem - g * T;
    const uint64_t key = a.S.seed + (uint64_t)trial;
    const int32_t* px = a.px + (size_t)trial * R * T;
    const int32_t* py = a.py + (size_t)trial * R * T;
    const int32_t* bx = a.bx + (size_t)trial * a.S.n_beacons;
    const int32_t* by = a.by + (size_t)trial * a.S.n_beacons;
    const int32_t* pz = a.S.dim == 3 ? a.pz + (size_t)trial * R * T : nullptr;
    const int32_t* bz = a.S.dim == 3 ? a.bz + (size_t)trial * a.S.n_beacons : nullptr;
    if (!FILL) a.cnt[i] = gen_ranges_at(a.S, key, g, t, px, py, bx, by, nullptr, nullptr, nullptr, nullptr);
    else {
        const int32_t o = a.off[i];
        gen_ranges_at(a.S, key, g, t, px, py, bx, by, a.ra + o, a.rb + o, a.dist + o, a.prec + o, pz, bz);
    }
}
#endif

}  // namespace score
