"""The lights' shadow tables (csrc/rt_shadowtab.inc) against brute-force rays.

rt_set_scene tabulates, per light, which objects a PCF shadow ray (raytracingCs.glsl:342-397) of a shading point can
possibly hit -- indexed by the direction and distance of the point as seen from a point / area light, or by its
projection on the plane orthogonal to a directional light -- and the packet kernel visits only those objects.  The
pixel-level gate is the bit-exact comparison with the oracle (tests/test_gpu_parity.py: every config, the fuzz scenes);
this file checks the TABLES themselves: for random shading points on the scene's surfaces, every object that any of the
point's jittered rays hits (shape test in fp64, with the shader's own hit conditions) must have its bit set in the cell the
kernel's fp32 lookup reads.  It also reports how selective the tables are.
"""
import numpy as np
import pytest

from opengl_raytracing_amd import layout as L
from opengl_raytracing_amd import scenes

pytestmark = pytest.mark.gpu

F = np.float32


def _halton(i, base):
    f, r = 1.0 / base, 0.0
    while i > 0:
        r += f * (i % base)
        i //= base
        f /= base
    return r


def _plane_basis(n):
    """right / forward of intersectPlane (:129-138) for one normal, float64."""
    up = np.array([0.0, 0.0, 1.0]) if abs(n[1]) > 0.9 else np.array([0.0, 1.0, 0.0])
    r = np.cross(n, up)
    r = r / np.linalg.norm(r)
    f = np.cross(r, n)
    f = f / np.linalg.norm(f)
    return r, f


def _lookup(hdr, cells, nw, O, Lpos, ldir, D, nn, n_obj, blocker=False):
    """st_lane_mask (rt_packet.inc) in fp32 for M lanes: -> uint32[M, nw]."""
    M = len(O)
    kind, base, K, NB = (int(hdr[0].view(np.int32)[k]) for k in range(4))
    if blocker:                      # the light's second table (pcssShadow's blocker rays): same cells, its own base
        base = int(hdr[6].view(np.int32)[0])
        if base == 0:
            kind = 0
    valid = np.zeros(nw, dtype=np.uint32)
    for k in range(nw):
        nb = n_obj - 32 * k
        valid[k] = 0xFFFFFFFF if nb >= 32 else ((1 << nb) - 1 if nb > 0 else 0)
    out = np.tile(valid, (M, 1))
    if kind == 0:
        return out
    if kind == 1:
        u = (O - Lpos).astype(F)
        a = np.abs(u)
        mx = (a[:, 0] >= a[:, 1]) & (a[:, 0] >= a[:, 2])
        my = ~mx & (a[:, 1] >= a[:, 2])
        ma = np.where(mx, a[:, 0], np.where(my, a[:, 1], a[:, 2]))
        um = np.where(mx, u[:, 0], np.where(my, u[:, 1], u[:, 2]))
        us = np.where(mx, u[:, 1], u[:, 0])
        ut = np.where(mx | my, u[:, 2], u[:, 1])
        face = np.where(mx, 0, np.where(my, 2, 4)) + (um < 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = (F(1.0) / ma).astype(F)
            half = F(0.5 * K)
            ia = np.floor((us * inv) * half + half)
            ib = np.floor((ut * inv) * half + half)
        ia = np.clip(np.nan_to_num(ia), 0, K - 1).astype(np.int64)
        ib = np.clip(np.nan_to_num(ib), 0, K - 1).astype(np.int64)
        ok = (D < hdr[1][1]) & (nn <= hdr[1][3])
        bins = np.clip((D * hdr[1][0]).astype(np.int64), 0, NB - 1)
        idx = ((bins * 6 + face) * K + ib) * K + ia
    else:
        T0, B0 = hdr[3][:3], hdr[4][:3]
        p = (O[:, 2] * T0[2] + O[:, 1] * T0[1]) + O[:, 0] * T0[0]
        q = (O[:, 2] * B0[2] + O[:, 1] * B0[1]) + O[:, 0] * B0[0]
        w = (O[:, 2] * ldir[2] + O[:, 1] * ldir[1]) + O[:, 0] * ldir[0]
        fp, fq, fb = (p - hdr[2][0]) * hdr[2][2], (q - hdr[2][1]) * hdr[2][2], (w - hdr[1][2]) * hdr[1][0]
        inside = (fp >= 0) & (fp < K) & (fq >= 0) & (fq < K)
        ip = np.clip(fp.astype(np.int64), 0, K - 1)
        iq = np.clip(fq.astype(np.int64), 0, K - 1)
        bins = np.clip(np.minimum(fb, 65536.0).astype(np.int64), 0, NB - 1)
        idx = np.where(inside, (bins * K + iq) * K + ip, NB * K * K)
        ok = (fb >= 0) & np.isfinite(fp) & np.isfinite(fq) & np.isfinite(fb)
    for k in range(nw):
        out[:, k] = np.where(ok, cells[base + idx * nw + k] & valid[k], valid[k])
    return out


def _check_scene(tracer, sc, rng, n_points=600):
    tracer.load(sc)
    tab, nw = tracer.shadow_tables()
    assert tab is not None, "rt_set_scene built no shadow tables"
    objs, lts = sc.objects, sc.lights
    n_obj, n_lt = len(objs), len(lts)
    hdrs = tab[: n_lt * 28].view(F).reshape(n_lt, 7, 4)
    # shading points: on the spheres and on the plane rectangles (finite records only), as the kernel produces them
    fin = np.isfinite(objs["position"]).all(axis=1) & np.isfinite(objs["radius"]) & np.isfinite(objs["normal"]).all(axis=1) & \
        np.isfinite(objs["size"]).all(axis=1) & (objs["type"] <= 1)
    ids = np.flatnonzero(fin)
    if len(ids) == 0:
        return None
    pick = rng.choice(ids, n_points)
    P = np.zeros((n_points, 3))
    N = np.zeros((n_points, 3))
    for k, i in enumerate(pick):
        o = objs[i]
        if o["type"] == 0:
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            P[k] = o["position"].astype(np.float64) + abs(float(o["radius"])) * d
            N[k] = d
        else:
            n = o["normal"].astype(np.float64)
            if not np.isfinite(n).all() or np.linalg.norm(np.cross(n, [0, 0, 1] if abs(n[1]) > 0.9 else [0, 1, 0])) == 0:
                P[k], N[k] = o["position"], (0, 1, 0)
                continue
            r, f = _plane_basis(n)
            P[k] = o["position"].astype(np.float64) + rng.uniform(-0.5, 0.5) * float(o["size"][0]) * r + rng.uniform(-0.5, 0.5) * float(o["size"][1]) * f
            N[k] = n
    P32, N32 = P.astype(F), N.astype(F)
    O32 = (P32 + N32 * F(0.001)).astype(F)
    nn = (N32[:, 2] * N32[:, 2] + N32[:, 1] * N32[:, 1]) + N32[:, 0] * N32[:, 0]
    stats = {"cells_mean_bits": [], "lanes_all": 0, "lanes": 0, "hits": 0}
    for li in range(n_lt):
        lt = lts[li]
        ltype = int(lt["type"])
        if ltype not in (0, 1, 2):
            continue
        fs = float(F(lt["shadowSoftness"]) * F(0.005))
        Lpos = lt["position"].astype(F)
        if ltype == 1:
            d = -lt["direction"].astype(np.float64)
            if not np.isfinite(d).all() or np.linalg.norm(d) == 0:
                continue
            ldir = np.tile((d / np.linalg.norm(d)), (n_points, 1))
            D = np.full(n_points, 1e6)
            limit = np.full(n_points, float(sc.params().maxRayDistance))
        else:
            raw = Lpos.astype(np.float64) - P32.astype(np.float64)
            D = np.linalg.norm(raw, axis=1)
            with np.errstate(invalid="ignore", divide="ignore"):
                ldir = raw / D[:, None]
            limit = np.minimum(float(sc.params().maxRayDistance), D)
        lane = _lookup(hdrs[li], tab, nw, O32, Lpos, ldir[0].astype(F), D.astype(F), nn.astype(F), n_obj)
        kind = int(hdrs[li][0].view(np.int32)[0])
        stats["lanes"] += n_points
        stats["lanes_all"] += int((lane == np.array([0xFFFFFFFF] * nw, dtype=np.uint32)).all(axis=1).sum()) if kind else n_points
        stats["cells_mean_bits"].append(float(np.mean([bin(int(x)).count("1") for row in lane for x in row]) * nw))
        # the lane's jittered rays (:352-375), fp64
        up = np.array([0.0, 1.0, 0.0])
        with np.errstate(invalid="ignore", divide="ignore"):
            T = np.cross(ldir, up)
            T /= np.linalg.norm(T, axis=1)[:, None]
            B = np.cross(ldir, T)
        ns = max(int(lt["pcfSamples"]), 1)
        for s in range(min(ns, 16)):
            rx, ry = _halton(s, 2) % 1.0, _halton(s, 3) % 1.0
            jd = ldir + T * (rx * fs) + B * (ry * fs)
            if ltype != 1:
                with np.errstate(invalid="ignore", divide="ignore"):
                    jd = jd / np.linalg.norm(jd, axis=1)[:, None]
            Oo = O32.astype(np.float64)
            for j in range(n_obj):
                o = objs[j]
                t = np.full(n_points, np.nan)
                if o["type"] == 0:
                    c, r = o["position"].astype(np.float64), float(o["radius"])
                    oc = Oo - c
                    a = (jd * jd).sum(1)
                    b = 2.0 * (oc * jd).sum(1)
                    cc = (oc * oc).sum(1) - r * r
                    disc = b * b - 4 * a * cc
                    with np.errstate(invalid="ignore", divide="ignore"):
                        t = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.nan)
                elif o["type"] == 1:
                    n = o["normal"].astype(np.float64)
                    if not np.isfinite(n).all() or not np.isfinite(o["position"]).all():
                        continue
                    upv = [0, 0, 1] if abs(n[1]) > 0.9 else [0, 1, 0]
                    if np.linalg.norm(np.cross(n, upv)) == 0:
                        continue
                    r_, f_ = _plane_basis(n)
                    denom = jd @ n
                    with np.errstate(invalid="ignore", divide="ignore"):
                        tt = ((o["position"].astype(np.float64) - Oo) @ n) / denom
                    lo = Oo + jd * tt[:, None] - o["position"].astype(np.float64)
                    okp = (np.abs(denom) > 1e-6) & (tt >= 0) & (np.abs(lo @ r_) <= float(o["size"][0]) / 2) & (np.abs(lo @ f_) <= float(o["size"][1]) / 2)
                    t = np.where(okp, tt, np.nan)
                else:
                    continue
                hit = (t > 0) & (t < limit)
                bit = (lane[:, j >> 5] >> np.uint32(j & 31)) & np.uint32(1)
                bad = hit & (bit == 0)
                stats["hits"] += int(hit.sum())
                assert not bad.any(), (f"{sc.name}: light {li} (type {ltype}, table kind {kind}) sample {s}: object {j} is hit by "
                                       f"{int(bad.sum())} shading point(s) whose table cell does not list it, e.g. P = {P[np.flatnonzero(bad)[0]]}")
        # pcssShadow's 16 blocker rays (:409-427) against the light's BLOCKER table
        if int(lt["shadowType"]) == 2 and int(hdrs[li][6].view(np.int32)[0]) != 0:
            blane = _lookup(hdrs[li], tab, nw, O32, Lpos, ldir[0].astype(F), D.astype(F), nn.astype(F), n_obj, blocker=True)
            stats["blocker_mean_bits"] = stats.get("blocker_mean_bits", []) + [float(np.mean([bin(int(x)).count("1") for row in blane for x in row]) * nw)]
            ss = float(F(lt["lightSize"]) * F(0.1))
            for sidx in range(16):
                rr = _halton(sidx, 3) * 2.0 - 1.0
                jd = ldir + rr * ss + rr * ss
                with np.errstate(invalid="ignore", divide="ignore"):
                    jd = jd / np.linalg.norm(jd, axis=1)[:, None]
                Oo = O32.astype(np.float64)
                for j in range(n_obj):
                    o = objs[j]
                    if o["type"] == 0:
                        c, r = o["position"].astype(np.float64), float(o["radius"])
                        oc = Oo - c
                        a = (jd * jd).sum(1)
                        b = 2.0 * (oc * jd).sum(1)
                        cc = (oc * oc).sum(1) - r * r
                        disc = b * b - 4 * a * cc
                        with np.errstate(invalid="ignore", divide="ignore"):
                            t = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.nan)
                    elif o["type"] == 1:
                        n = o["normal"].astype(np.float64)
                        if not np.isfinite(n).all() or not np.isfinite(o["position"]).all():
                            continue
                        upv = [0, 0, 1] if abs(n[1]) > 0.9 else [0, 1, 0]
                        if np.linalg.norm(np.cross(n, upv)) == 0:
                            continue
                        r_, f_ = _plane_basis(n)
                        denom = jd @ n
                        with np.errstate(invalid="ignore", divide="ignore"):
                            tt = ((o["position"].astype(np.float64) - Oo) @ n) / denom
                        lo = Oo + jd * tt[:, None] - o["position"].astype(np.float64)
                        okp = (np.abs(denom) > 1e-6) & (tt >= 0) & (np.abs(lo @ r_) <= float(o["size"][0]) / 2) & (np.abs(lo @ f_) <= float(o["size"][1]) / 2)
                        t = np.where(okp, tt, np.nan)
                    else:
                        continue
                    hit = (t > 0) & (t < limit)
                    bit = (blane[:, j >> 5] >> np.uint32(j & 31)) & np.uint32(1)
                    bad = hit & (bit == 0)
                    stats["blocker_hits"] = stats.get("blocker_hits", 0) + int(hit.sum())
                    assert not bad.any(), (f"{sc.name}: light {li} (type {ltype}) blocker ray {sidx}: object {j} is hit by {int(bad.sum())} "
                                           f"shading point(s) whose BLOCKER table cell does not list it, e.g. P = {P[np.flatnonzero(bad)[0]]}")
    return stats


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_tables_cover_every_hit_of_the_configs_scenes(tracer, host, cfg):
    sc = scenes.make_scene(cfg, host.generate_aabb)
    st = _check_scene(tracer, sc, np.random.default_rng(cfg), n_points=400 if cfg == 5 else 800)
    assert st["hits"] > 0
    print(f"C{cfg}: mean candidate bits per lane and light {np.mean(st['cells_mean_bits']):.2f} of {len(sc.objects)}; "
          f"{st['lanes_all']} of {st['lanes']} lanes take every object; {st['hits']} ray hits checked")
    # selective: a shading point keeps a small part of the scene per light
    assert np.mean(st["cells_mean_bits"]) < 0.5 * len(sc.objects)
    if cfg == 3:        # PCSS lights: the blocker tables were checked too
        assert st.get("blocker_hits", 0) > 0
        print(f"C3: mean candidate bits per lane and light in the BLOCKER tables {np.mean(st['blocker_mean_bits']):.2f} of {len(sc.objects)}; "
              f"{st['blocker_hits']} blocker-ray hits checked")


def test_tables_cover_every_hit_of_fuzzed_scenes(tracer, host):
    from test_gpu_parity import _fuzz_scene
    total = 0
    for seed in range(0, 60):
        sc = _fuzz_scene(seed)
        if len(sc.objects) > 256 or len(sc.lights) == 0:
            continue
        st = _check_scene(tracer, sc, np.random.default_rng(1000 + seed), n_points=150)
        if st:
            total += st["hits"]
    assert total > 0


def test_two_phase_build_equals_the_one_phase_build(host):
    """rt_set_scene builds the tables in two phases (every object once per DIRECTION cell, then the per-bin tests for the objects
    that survived: rt_shadowtab.inc); RT_ST_BUILD=full keeps the one-phase builder that tests every object in every cell.  The
    tests are monotone along the bin axis, so both must produce the same tables: headers identical, every cell of the two-phase
    table a subset of the one-phase cell (that alone keeps it rigorous only together with the monotonicity argument, hence:) and
    equal in all but a vanishing fraction of cells (fp32 borderline cases of the monotonicity)."""
    import os
    from test_gpu_parity import _fuzz_scene
    todo = [scenes.make_scene(cfg, host.generate_aabb) for cfg in (2, 3, 4, 5)]
    todo += [sc for sc in (_fuzz_scene(seed) for seed in range(0, 24)) if 0 < len(sc.objects) <= 256 and len(sc.lights) > 0]
    cells = diff = 0
    for sc in todo:
        with host.RayTracer(0) as two:
            two.load(sc)
            t2, nw = two.shadow_tables()
        os.environ["RT_ST_BUILD"] = "full"
        try:
            with host.RayTracer(0) as one:
                one.load(sc)
                t1, nw1 = one.shadow_tables()
        finally:
            del os.environ["RT_ST_BUILD"]
        if t2 is None:
            assert t1 is None
            continue
        assert nw == nw1 and t1.shape == t2.shape
        nh = len(sc.lights) * 7 * 4
        assert np.array_equal(t1[:nh], t2[:nh]), f"{sc.name}: headers differ"
        for li in range(len(sc.lights)):         # each light's cells: [base, base + nCells * nw) dwords (the rest of the buffer is unused)
            hdr = t1[li * 28:(li + 1) * 28]
            kind, base, n_cells = int(hdr[0]), int(hdr[1]), int(hdr[11])        # header ints are stored as raw bits
            if kind == 0:
                continue
            a, b = t1[base:base + n_cells * nw], t2[base:base + n_cells * nw]
            assert not (b & ~a).any(), f"{sc.name} light {li}: the two-phase table lists an object the one-phase table does not"
            cells += a.size
            diff += int((a != b).sum())
            bbase = int(hdr[24])             # the blocker table of a PCSS light (0: none)
            if bbase:
                a, b = t1[bbase:bbase + n_cells * nw], t2[bbase:bbase + n_cells * nw]
                assert not (b & ~a).any(), f"{sc.name} light {li}: the two-phase BLOCKER table lists an object the one-phase table does not"
                cells += a.size
                diff += int((a != b).sum())
    print(f"two-phase vs one-phase: {diff} of {cells} cell words differ")
    assert diff <= cells * 1e-5
