"""Bloom (brightness_extractFS / gaussian_blurFs / bloom_combineFs.glsl): oracle vs the
reference shaders chained on llvmpipe (fixture), HIP kernels vs the oracle through rt_bloom."""
import numpy as np
import pytest

from conftest import bits_equal, load_golden


def test_bloom_oracle_matches_reference_shaders(oracle):
    """Stage by stage.  The extract pass matches bit for bit on the power-of-two case (it pins the
    render-target rounding: fp32 -> fp16 toward zero).  Blur passes sample on texel centres; the rasteriser's TexCoords ulps
    leak ~1e-6 of a neighbouring texel into a tap, which can move a truncating fp16 store by one ulp:
    >= 99 % of pixels bit-exact per stage, the rest within one half-ulp (1/1024 relative) plus a
    neighbourhood-magnitude term, and the fp32 combine within 1.5e-3 relative."""
    g = load_golden("bloom")
    for tag in "ab":
        scene, stages, comb = g[f"{tag}_scene"], g[f"{tag}_stages"], g[f"{tag}_combined"]
        out, st = oracle.bloom(scene, 1.0, 0.5, 10, keep=True)
        assert len(st) == len(stages) == 11
        # extract: bit-exact on the power-of-two case (this pins the truncating store); on the ragged size the
        # TexCoords ulps move two texels across a truncation boundary, like in the blur passes
        e0 = (st[0].view(np.uint16) == stages[0].view(np.uint16)).all(axis=-1)
        assert e0.all() if tag == "a" else e0.mean() >= 0.998, f"case {tag} extract: exact {e0.mean():.4f}"
        a0, b0 = st[0].astype(np.float32), stages[0].astype(np.float32)
        assert (np.abs(a0 - b0) <= np.maximum(np.abs(a0), np.abs(b0)) / 1024 + 1e-7).all()
        for k in range(1, 11):
            a, b = st[k].astype(np.float32)[..., :3], stages[k].astype(np.float32)[..., :3]
            exact = (a == b).all(axis=-1).mean()
            assert exact >= 0.99, f"case {tag} stage {k}: exact {exact:.4f}"
            local = np.abs(stages[k - 1].astype(np.float32)[..., :3]).max()
            assert (np.abs(a - b) <= np.maximum(np.abs(a), np.abs(b)) / 1024 + 4e-6 * local + 1e-7).all(), f"case {tag} stage {k}"
            assert (st[k].view(np.uint16)[..., 3] == 0x3C00).all()
        err = np.abs(out - comb)
        assert (err <= 1.5e-3 * np.maximum(np.abs(out), np.abs(comb)) + 1e-6).all()
        assert (out == comb).all(axis=-1).mean() >= (0.9 if tag == "a" else 0.5)   # fp32 combine: ulp-level tap leakage on the ragged size


def test_bloom_threshold_and_identity(oracle):
    """Nothing above the threshold -> bloom texture stays zero and the combine returns the scene."""
    rng = np.random.default_rng(1)
    scene = rng.uniform(0, 0.3, (20, 30, 4)).astype(np.float32)
    out, st = oracle.bloom(scene, 1.0, 0.5, 10, keep=True)
    assert all((s.view(np.uint16)[..., :3] == 0).all() for s in st)
    assert np.array_equal(out[..., :3], scene[..., :3]) and (out[..., 3] == 1).all()
    scene[5, 7, :3] = (10.0, 10.0, 10.0)       # one bright texel spreads into a 5-iteration x 9-tap footprint
    out2, st2 = oracle.bloom(scene, 1.0, 0.5, 10, keep=True)
    last = st2[-1].astype(np.float32)[..., :3]
    assert last[5, 7].min() > 0 and last[5, 7 + 21:].max() == 0     # 5 horizontal passes x 4 taps = 20 texels
    assert 0.0 < float(last[..., 0].max()) < 10.0


@pytest.mark.gpu
def test_bloom_hip_bit_exact_vs_oracle(tracer, host, oracle):
    """rt_bloom against the oracle: fixture scenes, ragged and tiny sizes, 0 / odd iteration counts,
    HDR data with inf / NaN."""
    import torch
    g = load_golden("bloom")
    rng = np.random.default_rng(5)
    cases = [(g["a_scene"], 1.0, 0.5, 10), (g["b_scene"], 1.0, 0.5, 10), (g["a_scene"], 0.2, 1.5, 3), (g["b_scene"], 1.0, 0.5, 0)]
    cases += [(g["a_scene"], 0.5, 0.75, it) for it in (1, 2, 4, 7)]     # every fused/unfused pairing of the chain
    for (w, h) in [(1, 1), (3, 70), (129, 5), (200, 113), (64, 16), (65, 17)]:
        sc = (rng.uniform(0, 1, (h, w, 4)) ** 4 * 8).astype(np.float32)
        if w > 8:
            sc[h // 2, w // 2, 0] = np.inf
            sc[0, 1, 1] = np.nan
        cases.append((sc, 1.0, 0.5, 10))
    # fp16 round-toward-zero edge values through the extract store (the kernels convert with v_cvt_pkrtz_f16_f32,
    # the oracle in software): denormal halfs, the normal/denormal and overflow boundaries, both signs; green = 1
    # keeps every texel above the threshold and strength 2^40 makes the stored half recoverable from the output
    edge = np.array([0.0, 2.0 ** -25, 2.0 ** -24, 1.5 * 2.0 ** -24, 2.0 ** -24 * 1023.9, 2.0 ** -14, 2.0 ** -14 * (1 - 2.0 ** -12),
                     6.1e-5, 1.0 / 3.0, 0.1, 1.0009765, 1.00146, 65504.0, 65519.9, 65520.0, 65536.0, 1e5, 3e38, 1e-30, 1e-40],
                    dtype=np.float32)
    edge = np.concatenate([edge, -edge, rng.uniform(-7e4, 7e4, 88).astype(np.float32),
                           (rng.uniform(-1, 1, 128) * 2.0 ** rng.integers(-30, -10, 128)).astype(np.float32)])
    sc = np.zeros((16, 16, 4), np.float32)
    sc[..., 0] = edge.reshape(16, 16)
    sc[..., 1] = 1.0
    sc[..., 2] = edge.reshape(16, 16)[::-1, ::-1]
    cases += [(sc, 0.5, 2.0 ** 40, 0), (sc, 0.5, 2.0 ** 20, 2)]
    for scene, thr, strength, iters in cases:
        h, w = scene.shape[:2]
        want = oracle.bloom(scene, thr, strength, iters)
        d_scene = torch.from_numpy(np.ascontiguousarray(scene)).cuda()
        d_out = torch.empty_like(d_scene)
        tracer.bloom(d_scene.data_ptr(), d_out.data_ptr(), w, h, thr, strength, iters)
        tracer.sync()
        assert bits_equal(d_out.cpu().numpy(), want), f"{w}x{h} thr {thr} iters {iters}"
