"""Pins the oracle (oracle/rt_oracle.c, the CPU restatement) to the REFERENCE: outputs of
/root/reference/shader/raytracingCs.glsl executed unmodified on Mesa llvmpipe, committed as
fixtures by tests/golden/make_golden.py.  CPU only.

The oracle restates llvmpipe's own sin / cos / tan (tests/test_oracle_units.py pins them bitwise), so the camera
scale tan(radians(fov)*0.5), the per-depth bounce sample and the Russian-roulette hash are the reference's values
and NO test-only knob is involved any more.  Gates per fixture:

* gPosition / gNormal BIT FOR BIT (everything geometric is the same IEEE arithmetic in the same order);
* gColor within 1e-4 relative: the only arithmetic difference left is pow(x,5) -- exact product here and in the
  HIP kernel, exp2(5 log2 x) polynomials on llvmpipe (<= 1.3e-6 apart);
* DIAGNOSTIC: with llvmpipe's polynomial pow restated too (orc_params.reserved0 bit 0) gColor is BIT-EXACT as well on
  100 % of the pixels of C1..C4 and the NaN scene (C5, whose paths end in bilinear skybox taps: >= 90 %), which proves
  the claim above.
* The two residues of round 2 are EXPLAINED, both from the reference shader's final NIR as llvmpipe
  compiles it (LP_DEBUG=cs on the harness):
  - C3 (20 of 32 400 px, gPosition an ulp off after a diffuse bounce): cosineWeightedHemisphere's bitangent.x =
    n.y*t.z - n.z*t.y with t.y = -t.z is FACTORED by NIR into t.z * (n.y + n.z) -- one rounding fewer, invisible while
    the hemisphere sample's sin(phi) is ~0 (frameCount 0: C2, C4, C5), visible at C3's frameCount 7.  Restated (oracle
    and HIP): C3's gPosition is the reference's on 100 %.
  - C5 at MAX_RAY_DEPTH >= 8 (8 of 14 400 px, colour LOWER in the reference by percents, identical gPosition): gallivm
    gives a shader ONE counter of 65 535 loop iterations shared by all its loops (LP_MAX_TGSI_LOOP_ITERS, a hang guard);
    8 bounces x (1 + 8 lights x 4 samples) traversals x 257 passes of the 256-object loop = 67 848 exceeds it, so in
    pixels whose path is alive at the 8th bounce the reference-on-llvmpipe stops adding lights.  An artefact of the
    software rasteriser, not of the shader (a GPU driver has no such counter) -- class (iv), NOT restated in the product.
    test_c5_depth8_residue_is_llvmpipes_loop_limiter emulates the counter (reserved0 bit 1) and checks that it accounts
    for the residue.
"""
import numpy as np
import pytest

from conftest import GoldenScene, compare_surface, load_golden, params_from_bytes
from opengl_raytracing_amd import layout as L

# fixture -> (min exact fraction of gPosition/gNormal, min gColor pass fraction at 1e-4,
#             min gColor BIT-EXACT fraction with the polynomial-pow diagnostic)
GATES = {
    "c1": (1.0, 1.0, 1.0),
    "c2": (1.0, 1.0, 1.0),
    "c3": (1.0, 1.0, 1.0),
    "c4": (1.0, 1.0, 1.0),
    "c5": (1.0, 0.999, 0.90),
    "nan": (1.0, 1.0, 1.0),
}


def render_oracle(oracle, scene, params, mesa_pow=False):
    p = L.copy_params(params)
    p.reserved0 = 1 if mesa_pow else 0
    return oracle.render(scene, p)


@pytest.mark.parametrize("name", list(GATES))
def test_lowres_frame_against_reference(oracle, name):
    g = load_golden(name)
    sc = GoldenScene(g)
    p = params_from_bytes(g["lowres_params"])
    exact_min, color_min, color_exact_min = GATES[name]
    col, pos, nrm, rays = render_oracle(oracle, sc, p)
    cp = compare_surface(pos, g["lowres_pos"], rtol=0, atol=0)
    cn = compare_surface(nrm.astype(np.float32), g["lowres_normal"].astype(np.float32), rtol=0, atol=0)
    cc = compare_surface(col, g["lowres_color"])
    assert cp["exact_frac"] >= exact_min, f"gPosition exact {cp['exact_frac']:.6f}"
    assert cn["exact_frac"] >= exact_min, f"gNormal exact {cn['exact_frac']:.6f}"
    assert cc["pass_frac"] >= color_min, f"gColor pass {cc['pass_frac']:.6f} ({cc['n_fail']} px)"
    # NaN pixels (reference UB corners) must be NaN in both
    assert (np.isnan(col).any(axis=-1) == np.isnan(g["lowres_color"]).any(axis=-1)).mean() >= color_min
    # diagnostic: llvmpipe's polynomial pow restated as well -> gColor bit for bit
    col2, pos2, nrm2, _ = render_oracle(oracle, sc, p, mesa_pow=True)
    ce = compare_surface(col2, g["lowres_color"], rtol=0, atol=0)
    assert ce["exact_frac"] >= color_exact_min, f"gColor exact (polynomial pow) {ce['exact_frac']:.6f}"
    # miss pixels: (0,0,0,1) on all three surfaces
    miss = (g["lowres_pos"][..., :3] == 0).all(axis=-1) & (g["lowres_normal"][..., :3] == 0).all(axis=-1)
    if miss.any():
        assert (pos[miss][:, 3] == 1).all() and (col[miss][:, 3] == 1).all()
        assert (nrm[miss].astype(np.float32) == np.array([0, 0, 0, 1], dtype=np.float32)).all()


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5"])
def test_fullres_windows_against_reference(oracle, name):
    """Eight 32x32 windows cut from the reference's FULL-resolution frame (C5: its scene at
    1920x1080); the restatement renders just those windows."""
    g = load_golden(name)
    if "win_params" not in g.files:
        pytest.skip("no full-resolution windows in this fixture")
    sc = GoldenScene(g)
    base = params_from_bytes(g["win_params"])
    exact_min, color_same_min, _ = GATES[name]
    exact_min, color_same_min = min(exact_min, 0.9995), min(color_same_min, 0.9995)
    n_px = n_exact_p = n_exact_n = n_ok_c = 0
    for k, (x0, y0) in enumerate(g["win_origins"]):
        p = L.copy_params(base, x0=int(x0), y0=int(y0), regionW=32, regionH=32)
        col, pos, nrm, _ = render_oracle(oracle, sc, p)
        n_px += 32 * 32
        n_exact_p += compare_surface(pos, g["win_pos"][k], rtol=0, atol=0)["exact_mask"].sum()
        n_exact_n += compare_surface(nrm.astype(np.float32), g["win_normal"][k].astype(np.float32), rtol=0, atol=0)["exact_mask"].sum()
        n_ok_c += compare_surface(col, g["win_color"][k])["ok_mask"].sum()
    print(f"{name}: windows gPosition exact {n_exact_p / n_px:.6f} gNormal exact {n_exact_n / n_px:.6f} gColor pass {n_ok_c / n_px:.6f}")
    assert n_exact_p / n_px >= exact_min, f"gPosition exact {n_exact_p / n_px:.6f}"
    assert n_exact_n / n_px >= exact_min, f"gNormal exact {n_exact_n / n_px:.6f}"
    assert n_ok_c / n_px >= color_same_min, f"gColor pass {n_ok_c / n_px:.6f}"


def test_window_and_strip_renders_equal_full_frame(oracle):
    """Windows and interleaved strips are pure re-indexing: identical bits to the full frame."""
    g = load_golden("c2")
    sc = GoldenScene(g)
    p = params_from_bytes(g["lowres_params"])
    col, pos, nrm, rays = oracle.render(sc, p)
    w, h = p.width, p.height
    pw = L.copy_params(p, x0=37, y0=21, regionW=50, regionH=40)
    c2, p2, n2, _ = oracle.render(sc, pw)
    assert np.array_equal(c2, col[21:61, 37:87], equal_nan=True) and np.array_equal(p2, pos[21:61, 37:87], equal_nan=True)
    from opengl_raytracing_amd.dist import StripPlan
    plan = StripPlan(w, h, 16, 3)
    total = 0
    for r in range(3):
        ps = plan.params(p, r)
        cs, _, ns, rr = oracle.render(sc, ps)
        total += rr
        for ly in range(plan.max_local_rows):
            gy = plan.global_row(r, ly)
            if gy < h:
                assert np.array_equal(cs[ly], col[gy], equal_nan=True)
            else:
                assert (cs[ly] == 0).all() and (ns[ly].view(np.uint16) == 0).all()
    assert total == rays


def test_c5_at_7680x4320_windows_against_reference(oracle):
    """C5 at its REAL size ('parity unpinned at 8K'): 26 windows of 48x48 cut from the reference's
    7680x4320 frame -- partial dispatches of the unmodified GLSL on the full-size images (bottom 768 rows, and the
    left 128 columns over the whole height), so uv (imageSize) and random()'s arguments (gid up to 4319 + depth) are
    those of the 8K frame (raytracingCs.glsl:200-211, :273-275).  The oracle renders just those windows."""
    g = load_golden("c5_8k")
    sc = GoldenScene(dict(objects=g["objects"], lights=g["lights"], frame_count=g["frame_count"], has_noise=0, has_skybox=1))
    base = params_from_bytes(g["params"])
    assert (base.width, base.height) == (7680, 4320) and base.maxRayDepth == 8
    win = g["win8k_color"].shape[1]
    n_px = ep = en = okc = 0
    for k, (x0, y0) in enumerate(g["win8k_origins"]):
        p = L.copy_params(base, x0=int(x0), y0=int(y0), regionW=win, regionH=win)
        col, pos, nrm, _ = oracle.render(sc, p)
        n_px += win * win
        ep += compare_surface(pos, g["win8k_pos"][k], rtol=0, atol=0)["exact_mask"].sum()
        en += compare_surface(nrm.astype(np.float32), g["win8k_normal"][k].astype(np.float32), rtol=0, atol=0)["exact_mask"].sum()
        okc += compare_surface(col, g["win8k_color"][k])["ok_mask"].sum()
    print(f"c5 8K: gPosition exact {ep / n_px:.6f} gNormal exact {en / n_px:.6f} gColor pass {okc / n_px:.6f} ({n_px} px)")
    assert ep / n_px >= 0.9995 and en / n_px >= 0.9995
    assert okc / n_px >= 0.999


def test_c5_depth8_residue_is_llvmpipes_loop_limiter(oracle):
    """C5's colour residue at MAX_RAY_DEPTH >= 8 (module docstring) is llvmpipe's 65 535-iteration loop limiter:
    (a) it needs depth >= 8 AND 256 objects AND 8 shadowed lights -- 8 x 33 x 257 > 65 535 >= 7 x 33 x 257 + 257;
    (b) with the counter emulated per pixel (reserved0 bit 1) five of the eight pixels fall within 1e-4 of the reference and two
        of them reproduce it to the bit pattern of their position in the frame (llvmpipe counts per 8-invocation VECTOR: a
        neighbouring lane on a subsurface material spends budget for the whole vector -- bit 2 emulates that extreme; the
        remaining pixels' reference colours lie BETWEEN the two emulations);
    (c) every pixel outside that set is untouched by the emulation, and at depth 7 nothing differs at all."""
    assert 8 * 33 * 257 > 65535 >= 7 * 33 * 257 + 257
    g = load_golden("c5")
    sc = GoldenScene(g)
    p = params_from_bytes(g["lowres_params"])
    assert p.maxRayDepth == 8 and len(sc.objects) == 256 and len(sc.lights) == 8
    ref = g["lowres_color"]
    cols = {}
    for flags in (0, 2, 6):
        q = L.copy_params(p)
        q.reserved0 = flags
        cols[flags] = oracle.render(sc, q)[0]
    fail = {f: ~compare_surface(cols[f], ref)["ok_mask"] for f in cols}
    assert int(fail[0].sum()) == 8
    assert int((fail[0] & ~fail[2]).sum()) >= 5, "the per-pixel limiter must bring at least five residue pixels within 1e-4"
    # the pixels neither variant brings within 1e-4: the reference lies between the two emulations (per-vector counting)
    stubborn = fail[0] & fail[2] & fail[6]
    lum = lambda a: a[..., :3].sum(-1)
    lo, hi = np.minimum(lum(cols[2]), lum(cols[6])), np.maximum(lum(cols[2]), lum(cols[6]))
    assert ((lum(ref)[stubborn] >= lo[stubborn] * (1 - 1e-4)) & (lum(ref)[stubborn] <= hi[stubborn] * (1 + 1e-4))).all()
    # the emulation touches only pixels whose path is alive at the 8th bounce (a handful), never the rest of the frame
    assert int((cols[2] != cols[0]).any(-1).sum()) <= 40
    # and nothing of this exists at depth 7
    q7 = L.copy_params(p, maxRayDepth=7)
    c7 = oracle.render(sc, q7)[0]
    q7.reserved0 = 2
    assert np.array_equal(oracle.render(sc, q7)[0], c7, equal_nan=True)
