"""TEST INFRASTRUCTURE ONLY (the product never imports this): an independent restatement of the batched generator of
synthetic Manhattan worlds (score_amd/csrc/score_generate.hpp) in plain Python -- pure-Python loops, small cases only.

What is restated: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11: the published round
function and constants), the generator's use of it (key = seed + trial, counter = (purpose, robot / group, step, index)),
the lattice walk (81 / 9 / 9 / 1 % straight / left / right / back in weighted random order until the next step stays on the
grid), odometry and range measurements with Box-Muller noise.  The statistics it encodes are SURVEY 8(d)'s, measured from the
reference's shipped fixture /root/reference/examples/manhattan/factor_graph.pickle (tests/golden/manhattan_fg.npz); the
Philox known-answer vectors of Random123 pin the generator of random bits itself (tests/test_generate.py)."""
import math

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF
START, TURN, ODOM_A, ODOM_B, BEACON, HIT_RB, NOISE_RB, HIT_RR, NOISE_RR = 1, 2, 3, 4, 5, 6, 7, 8, 9


def philox4x32_10(key, c0, c1, c2, c3):
    k0, k1 = key & MASK, (key >> 32) & MASK
    x0, x1, x2, x3 = c0 & MASK, c1 & MASK, c2 & MASK, c3 & MASK
    for _ in range(10):
        p0, p1 = M0 * x0, M1 * x2
        x0, x1, x2, x3 = ((p1 >> 32) ^ x1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ x3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return x0, x1, x2, x3


def u01(hi, lo):
    return float(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0)


def normal2(p):
    u, v = 1.0 - u01(p[0], p[1]), u01(p[2], p[3])
    r, a = math.sqrt(-2.0 * math.log(u)), 2.0 * math.pi * v
    return r * math.cos(a), r * math.sin(a)


DIRS = ((1, 0), (0, 1), (-1, 0), (0, -1))


def inside(x, y, side):
    return 0 <= x <= side and 0 <= y <= side


def world(seed, trial, n_robots, n_poses, n_beacons, side=20, p_range=0.10, sigma_t=0.01, sigma_theta=0.002, sigma_range=1.0):
    """One world: dict with positions / headings per robot, beacons, odometry (base, to, (x, y), theta) and ranges
    (a, b, dist) in the generator's order."""
    key = (seed + trial) & 0xFFFFFFFFFFFFFFFF
    pos, hd, odom = [], [], []
    for r in range(n_robots):
        if r == 0:
            x, y, h = 0, 0, 0
        else:
            for attempt in range(256):
                p = philox4x32_10(key, START, r, attempt, 0)
                x, y, h = (p[0] * (side + 1)) >> 32, (p[1] * (side + 1)) >> 32, p[2] >> 30
                if inside(x + DIRS[h][0], y + DIRS[h][1], side):
                    break
        px, ph = [(x, y)], [h]
        for i in range(1, n_poses):
            x, y = x + DIRS[h][0], y + DIRS[h][1]
            p = philox4x32_10(key, TURN, r, i, 0)
            w, turn, tot = [0.81, 0.09, 0.09, 0.01], (0, 1, 3, 2), 1.0
            hn = (h + 2) & 3
            for draw in range(4):
                u = (p[draw] / 4294967296.0) * tot
                acc, pick = 0.0, -1
                for k in range(4):
                    if w[k] <= 0.0:
                        continue
                    pick = k
                    acc += w[k]
                    if u < acc:
                        break
                cand = (h + turn[pick]) & 3
                if inside(x + DIRS[cand][0], y + DIRS[cand][1], side):
                    hn = cand
                    break
                tot -= w[pick]
                w[pick] = 0.0
            e = i - 1
            dth = (((hn - h + 1) & 3) - 1) * (math.pi / 2)
            n0, n1 = normal2(philox4x32_10(key, ODOM_A, r, e, 0))
            n2, _ = normal2(philox4x32_10(key, ODOM_B, r, e, 0))
            th = dth + sigma_theta * n2
            th = math.atan2(math.sin(th), math.cos(th))
            odom.append((r * n_poses + e, r * n_poses + e + 1, 1.0 + sigma_t * n0, sigma_t * n1, th))
            h = hn
            px.append((x, y)); ph.append(h)
        pos.append(px); hd.append(ph)
    beacons = []
    for b in range(n_beacons):
        p = philox4x32_10(key, BEACON, b, 0, 0)
        beacons.append(((p[0] * (side + 1)) >> 32, (p[1] * (side + 1)) >> 32))
    ranges = []
    Np = n_robots * n_poses
    for r in range(n_robots):
        for t in range(n_poses):
            for b in range(n_beacons):
                p = philox4x32_10(key, HIT_RB, r, t, b >> 2)
                if p[b & 3] / 4294967296.0 < p_range:
                    n0, _ = normal2(philox4x32_10(key, NOISE_RB, r, t, b))
                    dx, dy = pos[r][t][0] - beacons[b][0], pos[r][t][1] - beacons[b][1]
                    ranges.append((r * n_poses + t, Np + b, max(0.0, math.sqrt(float(dx * dx + dy * dy)) + sigma_range * n0)))
    g = n_robots
    for a in range(n_robots):
        for b in range(a + 1, n_robots):
            for t in range(n_poses):
                p = philox4x32_10(key, HIT_RR, g, t, 0)
                if p[0] / 4294967296.0 < p_range:
                    n0, _ = normal2(philox4x32_10(key, NOISE_RR, g, t, 0))
                    dx, dy = pos[a][t][0] - pos[b][t][0], pos[a][t][1] - pos[b][t][1]
                    ranges.append((a * n_poses + t, b * n_poses + t, max(0.0, math.sqrt(float(dx * dx + dy * dy)) + sigma_range * n0)))
            g += 1
    return dict(pos=pos, hd=hd, beacons=beacons, odom=odom, ranges=ranges)


# ---------------------------------------------------------------------------------------------------------------------
# 3-D worlds (round 6): the counterpart of `world` on the lattice of the cube [0, side]^3, restated from the description in
# score_amd/csrc/score_generate.hpp.  An orientation is (forward axis, up axis), axes coded 0..5 = +x -x +y -y +z -z; the body
# frame is (forward, left = up x forward, up).
# ---------------------------------------------------------------------------------------------------------------------
TURN_B, ODOM_C, ODOM_D = 10, 11, 12
AXES = ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def rot3(o):
    """Rotation matrix (rows) of orientation o = 6 f + u: columns forward, left, up."""
    f, u = AXES[o // 6], AXES[o % 6]
    l = _cross(u, f)
    return [[f[k], l[k], u[k]] for k in range(3)]


def turned3(o, t):
    f, u = o // 6, o % 6
    l = AXES.index(_cross(AXES[u], AXES[f]))
    nf = (f, l, l ^ 1, u, u ^ 1, f ^ 1)[t]
    nu = (f ^ 1) if t == 3 else f if t == 4 else u
    return 6 * nf + nu


def inside3(p, side):
    return all(0 <= c <= side for c in p)


def _step(p, o):
    return tuple(p[k] + AXES[o // 6][k] for k in range(3))


def exp3(w):
    th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
    th = math.sqrt(th2)
    if th < 1e-8:
        a, b = 1.0 - th2 / 6.0, 0.5 - th2 / 24.0
    else:
        a, b = math.sin(th) / th, (1.0 - math.cos(th)) / th2
    x, y, z = w
    return [[1.0 - b * (y * y + z * z), -a * z + b * x * y, a * y + b * x * z],
            [a * z + b * x * y, 1.0 - b * (x * x + z * z), -a * x + b * y * z],
            [-a * y + b * x * z, a * x + b * y * z, 1.0 - b * (x * x + y * y)]]


def world3(seed, trial, n_robots, n_poses, n_beacons, side=20, p_range=0.10, sigma_t=0.01, sigma_theta=0.002, sigma_range=1.0):
    """One 3-D world: positions / orientation codes per robot, beacons, odometry (base, to, t (3), R (3 x 3)) and ranges."""
    key = (seed + trial) & 0xFFFFFFFFFFFFFFFF
    pos, ori, odom = [], [], []
    for r in range(n_robots):
        if r == 0:
            p, o = (0, 0, 0), 4
        else:
            for attempt in range(256):
                q = philox4x32_10(key, START, r, attempt, 0)
                p = tuple((q[k] * (side + 1)) >> 32 for k in range(3))
                f = ((q[3] & 0xFFFF0000) * 6) >> 32
                k = (((q[3] << 16) & MASK) * 4) >> 32
                u = [c for c in range(6) if (c >> 1) != (f >> 1)][k]
                o = 6 * f + u
                if inside3(_step(p, o), side):
                    break
        pp, oo = [p], [o]
        for i in range(1, n_poses):
            p = _step(p, o)
            qa, qb = philox4x32_10(key, TURN, r, i, 0), philox4x32_10(key, TURN_B, r, i, 0)
            bits = list(qa) + list(qb[:2])
            w, tot = [0.80, 0.045, 0.045, 0.045, 0.045, 0.02], 1.0
            on = turned3(o, 5)
            for draw in range(6):
                u_ = (bits[draw] / 4294967296.0) * tot
                acc, pick = 0.0, -1
                for k in range(6):
                    if w[k] <= 0.0:
                        continue
                    pick = k
                    acc += w[k]
                    if u_ < acc:
                        break
                cand = turned3(o, pick)
                if inside3(_step(p, cand), side):
                    on = cand
                    break
                tot -= w[pick]
                w[pick] = 0.0
            e = i - 1
            n0, n1 = normal2(philox4x32_10(key, ODOM_A, r, e, 0))
            n2, _ = normal2(philox4x32_10(key, ODOM_B, r, e, 0))
            n4, n5 = normal2(philox4x32_10(key, ODOM_C, r, e, 0))
            n6, _ = normal2(philox4x32_10(key, ODOM_D, r, e, 0))
            Ri, Rn, E = rot3(o), rot3(on), exp3((sigma_theta * n4, sigma_theta * n5, sigma_theta * n6))
            turn = [[sum(Ri[m][a] * Rn[m][k] for m in range(3)) for k in range(3)] for a in range(3)]
            Rm = [[sum(turn[a][k] * E[k][b] for k in range(3)) for b in range(3)] for a in range(3)]
            odom.append((r * n_poses + e, r * n_poses + e + 1, (1.0 + sigma_t * n0, sigma_t * n1, sigma_t * n2), Rm))
            o = on
            pp.append(p); oo.append(o)
        pos.append(pp); ori.append(oo)
    beacons = []
    for b in range(n_beacons):
        q = philox4x32_10(key, BEACON, b, 0, 0)
        beacons.append(tuple((q[k] * (side + 1)) >> 32 for k in range(3)))
    ranges = []
    Np = n_robots * n_poses

    def dist3(a, b):
        return math.sqrt(float(sum((a[k] - b[k]) ** 2 for k in range(3))))

    for r in range(n_robots):
        for t in range(n_poses):
            for b in range(n_beacons):
                q = philox4x32_10(key, HIT_RB, r, t, b >> 2)
                if q[b & 3] / 4294967296.0 < p_range:
                    n0, _ = normal2(philox4x32_10(key, NOISE_RB, r, t, b))
                    ranges.append((r * n_poses + t, Np + b, max(0.0, dist3(pos[r][t], beacons[b]) + sigma_range * n0)))
    g = n_robots
    for a in range(n_robots):
        for b in range(a + 1, n_robots):
            for t in range(n_poses):
                q = philox4x32_10(key, HIT_RR, g, t, 0)
                if q[0] / 4294967296.0 < p_range:
                    n0, _ = normal2(philox4x32_10(key, NOISE_RR, g, t, 0))
                    ranges.append((a * n_poses + t, b * n_poses + t, max(0.0, dist3(pos[a][t], pos[b][t]) + sigma_range * n0)))
            g += 1
    return dict(pos=pos, ori=ori, beacons=beacons, odom=odom, ranges=ranges)
