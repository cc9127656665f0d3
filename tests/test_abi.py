"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/nbc.h declares, describes the same network as the oracle, and packs a state_dict the
way the kernels expect.  No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from neuralbarkcalculator_amd import _lib, synth, topology
from neuralbarkcalculator_amd.model import FCNResNet50, pack_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nbc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(built_lib, s), f"libnbc_hip.so does not export {s}"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signature table and nbc.h disagree"
    assert b"gfx950" in built_lib.nbc_version()


def test_library_holds_no_measurement_switches(built_lib):
    """The timing-only variants of the conv kernel (no MFMA, no refill DMA, ...) and the A/B instruction switch
    exist in the -DNBC_DIAG builds of tools/ only: the product library neither reads their environment variables
    nor contains those kernel instantiations."""
    blob = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"NBC_CONV_ABLATE", b"NBC_CONV_MFMA32", b"getenv"):
        assert needle not in blob, needle


def test_topology_through_the_abi(built_lib):
    units = topology.conv_units()
    assert built_lib.nbc_num_convs() == len(units) == 55
    d = _lib.NbcConvDesc()
    for i, u in enumerate(units):
        assert built_lib.nbc_conv_info(i, C.byref(d)) == 0
        got = (d.name.decode(), d.bn.decode() or None, d.cin, d.cout, d.k, d.stride, d.pad, d.dil,
               bool(d.relu), bool(d.bias), bool(d.residual))
        want = (u.name, u.bn, u.cin, u.cout, u.k, u.stride, u.pad, u.dil, u.relu, u.bias, u.residual)
        assert got == want
    assert built_lib.nbc_conv_info(99, C.byref(d)) == _lib.NBC_ERR_INVALID
    assert b"bad index" in built_lib.nbc_last_error()


def test_state_keys_match_oracle(built_lib):
    from oracle.fcn_resnet50_oracle import OracleFCNResNet50
    sd = OracleFCNResNet50().state_dict()
    n = built_lib.nbc_num_state_keys()
    assert n == 326
    name = C.c_char_p()
    shape = (C.c_int64 * 4)()
    ndim, dtype = C.c_int32(), C.c_int32()
    keys = []
    for i in range(n):
        assert built_lib.nbc_state_key(i, C.byref(name), C.byref(shape), C.byref(ndim), C.byref(dtype)) == 0
        k = name.value.decode()
        keys.append(k)
        assert tuple(shape[: ndim.value]) == tuple(sd[k].shape), k
        assert dtype.value == (1 if sd[k].dtype == torch.int64 else 0)
    assert keys == list(sd.keys())


def test_lowres_size(built_lib):
    h, w = C.c_int(), C.c_int()
    for H, W in [(1024, 1024), (520, 1024), (203, 1024), (8, 8), (9, 17)]:
        assert built_lib.nbc_lowres_size(H, W, C.byref(h), C.byref(w)) == 0
        assert (h.value, w.value) == topology.out_hw(H, W)


def test_strict_key_check_like_load_state_dict(built_lib, sd_np):
    bad = dict(sd_np)
    bad.pop("backbone.layer2.0.conv1.weight")
    bad["classifier.5.weight"] = np.zeros((3, 3), np.float32)
    with pytest.raises(RuntimeError) as e:
        pack_state_dict(bad, "fp32")
    msg = str(e.value)
    assert "Missing key(s)" in msg and "backbone.layer2.0.conv1.weight" in msg
    assert "Unexpected key(s)" in msg and "classifier.5.weight" in msg
    wrong = dict(sd_np)
    wrong["classifier.4.weight"] = np.zeros((2, 512, 1, 1), np.float32)   # a "2-class head" is not this model
    with pytest.raises(RuntimeError, match="mismatch"):
        pack_state_dict(wrong, "bf16")


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _join_f16x2(raw_u8, row_elems, group):
    """A packed f16x2 row -> (h0, h1) as float32 arrays in element order: groups of `group` elements are stored as
    [h0 x group][h1 x group] (group = 32 channels, or 4 for the stem's 16-byte tap)."""
    h = raw_u8.view(np.float16).reshape(-1, row_elems // group, 2, group)
    return (h[:, :, 0, :].reshape(-1, row_elems).astype(np.float32), h[:, :, 1, :].reshape(-1, row_elems).astype(np.float32))


def test_split_f16x2_is_numpy_float16_rounding(built_lib):
    """nbc_split_f16x2 (the host side of NBC_PREC_F16X2): h0 = float16(x) rounded to nearest even with subnormals,
    overflow to infinity from 65520 on; h1 = float16((x - h0) * 2^11).  Against numpy's binary16 on magnitudes from
    subnormal to overflow, special values and 300 000 random bit patterns; and h0 + h1 / 2^11 gives x back to 2^-24."""
    rng = np.random.default_rng(0)
    parts = [rng.standard_normal(100000).astype(np.float32) * np.float32(s) for s in (1, 1e-3, 1e-5, 1e-7, 100, 6e4, 1e5)]
    parts.append(np.array([0, -0.0, 65504, 65519.996, 65520, 65536, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 2.0 ** -14,
                           np.inf, -np.inf, np.nan, 6.1e-5, 6.0975e-5, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11], dtype=np.float32))
    parts.append(rng.integers(0, 2 ** 32, 300000, dtype=np.uint64).astype(np.uint32).view(np.float32))
    x = np.concatenate(parts)
    h0 = np.zeros(x.size, np.uint16)
    h1 = np.zeros(x.size, np.uint16)
    assert built_lib.nbc_split_f16x2(x.ctypes.data, x.size, h0.ctypes.data, h1.ctypes.data) == 0
    with np.errstate(all="ignore"):
        w0 = x.astype(np.float16)
        w1 = ((x - w0.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    for got, want in ((h0, w0), (h1, w1)):
        same = (want.view(np.uint16) == got) | (np.isnan(want) & np.isnan(got.view(np.float16)))
        assert same.all(), x[~same][:5]
    normal = np.isfinite(x) & (np.abs(x) < 65504) & (np.abs(x) > 1e-3)
    back = h0.view(np.float16).astype(np.float64) + h1.view(np.float16).astype(np.float64) / 2048.0
    rel = np.abs(back[normal] - x[normal].astype(np.float64)) / np.abs(x[normal].astype(np.float64))
    assert rel.max() <= 2.0 ** -23, rel.max()


@pytest.mark.parametrize("precision", ["fp32", "bf16", "f16x2"])
def test_pack_layout_and_bn_fold(built_lib, sd_np, precision):
    blob = pack_state_dict(sd_np, precision)
    prec = {"fp32": 0, "bf16": 1, "f16x2": 2}[precision]
    assert blob.nbytes == built_lib.nbc_packed_weights_bytes(prec)
    eb = 2 if prec == 1 else 4
    off = 0

    def align(v):
        return (v + 255) // 256 * 256

    for u in topology.conv_units():
        w = sd_np[u.name + ".weight"]
        if u.bn is None:
            got = blob[off: off + w.size * 4].view(np.float32).reshape(3, 512)
            np.testing.assert_array_equal(got, w.reshape(3, 512))
            off = align(off + w.size * 4)
            np.testing.assert_array_equal(blob[off: off + 12].view(np.float32), sd_np[u.name + ".bias"])
            off = align(off + 12)
            continue
        if u.cin == 3:
            cin_pad, ksteps = 16 // eb, 7
        else:
            cin_pad, ksteps = u.cin, u.k * u.k * u.cin * eb // 128
        row = ksteps * 128 // eb
        raw = blob[off: off + u.cout * ksteps * 128]
        if prec == 2:        # two f16 pieces per element in the f32 mode's geometry
            g0, g1 = _join_f16x2(raw, row, 4 if u.cin == 3 else 32)
        else:
            got = raw.view(np.float32) if prec == 0 else _bf16_to_f32(raw.view(np.uint16))
            got = got.reshape(u.cout, row)
        want = np.zeros((u.cout, row), np.float32)
        khkwci = w.transpose(0, 2, 3, 1)                       # [O][kh][kw][I]
        for tap in range(u.k * u.k):
            # the stem keeps one kernel ROW per 128-byte K-step (eight 16-byte slots, kw of them used), so that a
            # K-step of an output pixel reads consecutive input pixels; every other conv is [kh][kw][Cin]
            slot = (tap // u.k) * 8 + tap % u.k if u.cin == 3 else tap
            want[:, slot * cin_pad: slot * cin_pad + u.cin] = khkwci[:, tap // u.k, tap % u.k, :]
        if prec == 1:
            want = torch.from_numpy(want).to(torch.bfloat16).float().numpy()   # RNE like the packer
        row_pow2 = np.ones(u.cout, np.float32)
        if prec == 2:
            # every output channel's row is normalised by the power of two that puts its largest |w| into [2^14, 2^15)
            # (exact) and split into P = f16(v), Q = f16(v - P) (Q not scaled, unlike an activation's low piece); the
            # inverse power goes into that channel's BatchNorm scale below
            row_pow2 = _row_pow2(want)
            want = want * row_pow2[:, None]
            assert (np.abs(want).max(1) >= 2.0 ** 14).all() and (np.abs(want).max(1) < 2.0 ** 15).all()
            w0 = want.astype(np.float16).astype(np.float32)
            np.testing.assert_array_equal(g0, w0, err_msg=u.name)
            np.testing.assert_array_equal(g1, (want - w0).astype(np.float16).astype(np.float32), err_msg=u.name)
        else:
            np.testing.assert_array_equal(got, want, err_msg=u.name)
        off = align(off + u.cout * ksteps * 128)
        scale = blob[off: off + u.cout * 4].view(np.float32)
        off = align(off + u.cout * 4)
        shift = blob[off: off + u.cout * 4].view(np.float32)
        off = align(off + u.cout * 4)
        g, b = sd_np[u.bn + ".weight"], sd_np[u.bn + ".bias"]
        mu, var = sd_np[u.bn + ".running_mean"], sd_np[u.bn + ".running_var"]
        inv = np.float32(1.0) / np.sqrt(var + np.float32(1e-5), dtype=np.float32)
        np.testing.assert_array_equal(scale, g * inv / row_pow2)            # a power of two: the division is exact
        np.testing.assert_array_equal(shift, b - mu * (g * inv))
    # the trailer: magic, NBC_PACK_* flags (none for this checkpoint), the unit count, one activation power per unit -- all
    # zero for a checkpoint of ordinary magnitudes, in every precision: the blob computes what round 4's computed
    meta = blob[off: off + 1024].view(np.int32)
    assert meta[0] == 0x4e424335 and meta[1] == 0 and meta[2] == len(topology.conv_units())
    assert not meta[3:].any()
    assert built_lib.nbc_packed_weights_flags(blob.ctypes.data, blob.nbytes, prec) == 0
    off = align(off + 1024)
    assert off == blob.nbytes


def _row_pow2(rows):
    """2^k per row with max |w| * 2^k in [2^14, 2^15) (1 for a row of zeros): what nbc_pack_weights normalises f16x2 rows by."""
    m = np.abs(rows).max(1).astype(np.float64)
    k = np.where(m > 0, 14 - np.floor(np.log2(np.where(m > 0, m, 1.0))), 0.0)
    return np.exp2(k).astype(np.float32)


@pytest.mark.parametrize("log2_scale", [-20, -16, -12, 0, 10, 30])
def test_f16x2_weight_rows_are_normalised_before_the_split(built_lib, sd_np, log2_scale):
    """The f16 pieces of a weight hold it to 2^-23 only while both are normal f16 numbers.  nbc_pack_weights multiplies
    every output channel's row by a power of two first (largest |w| into [2^14, 2^15), the top of f16's range) and the
    channel's f32 BatchNorm scale by the inverse, so a checkpoint whose convolution weights are 1e-6 (or 1e7) in
    magnitude -- a convolution in front of a BatchNorm is scale-free -- packs to the same pieces as the same network at
    ordinary magnitudes, and (P + Q) * 2^-k gives every weight back to 2^-23 of the row's largest."""
    name, bn = "backbone.layer3.2.conv2", "backbone.layer3.2.bn2"
    s = np.float32(2.0 ** log2_scale)
    sd = dict(sd_np)
    sd[name + ".weight"] = sd_np[name + ".weight"] * s
    blob, ref = pack_state_dict(sd, "f16x2"), pack_state_dict(sd_np, "f16x2")
    off = 0
    for u in topology.conv_units():
        if u.bn is None:
            break
        ksteps = 7 if u.cin == 3 else u.k * u.k * u.cin * 4 // 128
        wbytes = u.cout * ksteps * 128
        a = lambda v: (v + 255) // 256 * 256
        s_off = a(off + wbytes)
        if u.name == name:
            np.testing.assert_array_equal(blob[off: off + wbytes], ref[off: off + wbytes])        # the same pieces
            scale = blob[s_off: s_off + u.cout * 4].view(np.float32)
            scale_ref = ref[s_off: s_off + u.cout * 4].view(np.float32)
            np.testing.assert_array_equal(scale, scale_ref * s)                                    # the scale carries 2^-k
            g0, g1 = _join_f16x2(blob[off: off + wbytes], ksteps * 32, 32)
            w = sd[name + ".weight"].transpose(0, 2, 3, 1).reshape(u.cout, -1).astype(np.float64)
            inv = np.float32(1.0) / np.sqrt(sd[bn + ".running_var"] + np.float32(1e-5), dtype=np.float32)
            k = 1.0 / _row_pow2(w).astype(np.float64)                                              # 2^-k per row
            np.testing.assert_array_equal(scale, (sd[bn + ".weight"] * inv) * k.astype(np.float32))
            back = (g0.astype(np.float64) + g1.astype(np.float64)) * k[:, None]
            err = np.abs(back - w).max(1) / np.abs(w).max(1)
            assert err.max() <= 2.0 ** -23, err.max()
        off = a(a(s_off + u.cout * 4) + u.cout * 4)


def test_model_fails_loudly_without_gpu(built_lib, sd_np):
    m = FCNResNet50("fp32")
    assert m.eval() is m
    with pytest.raises(RuntimeError):
        m.to("cpu")                                 # the CPU path is the reference, not this package
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            m.to("cuda:0")                          # nbc_create reports the missing device
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 16, 16))                # no context / weights
    with pytest.raises(RuntimeError):
        FCNResNet50("fp32").train(True)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libnbc_hip.so"))
    with pytest.raises(RuntimeError, match="only implementation"):
        FCNResNet50("fp32")


def test_default_conv_tile_cost_model(built_lib):
    """Host logic of the per-layer default tile (csrc/conv_igemm_dma.hip, choose_conv_tile): a valid tile for the
    precision and channel count, and the choices that matter most, where whole rounds of blocks on 256 CUs decide."""
    rows = [128, 128, 256, 256, 128, 128, 256, 128, 64, 128, 128, 256, 256, 128, 128, 128, 128, 128]
    cols = [64, 128, 128, 256, 128, 256, 64, 64, 128, 128, 64, 128, 256, 128, 128, 64, 128, 128]
    f = built_lib.nbc_default_conv_tile
    for prec in (0, 1, 2):
        for co in (64, 128, 256, 512, 1024, 2048):
            for m in (64, 1000, 8448, 9984, 16384, 65536, 131072, 524288):
                for k in (64, 576, 2048, 18432):
                    t = f(m, co, k, prec)
                    assert 0 <= t < 18 and co % cols[t] == 0
                    assert not (prec == 0 and t in (3, 12))            # f32 has no 256x256 tile
                    assert prec != 2 or t not in (2, 3, 4, 11, 12)       # f16x2: wave tiles of 64x64 at most, no 16-wave blocks
                    assert prec == 2 or t < 14                             # the loader-wave and two-blocks-per-CU tiles are f16x2's
    assert f(16384, 96, 64, 0) == -1 and f(16384, 512, 64, 7) == -1 and f(0, 512, 64, 0) == -1
    blocks = lambda t, m, co: -(-m // rows[t]) * (co // cols[t])
    # the head conv (3x3, 2048 -> 512) in f32 at 1024x1024: 16 384 pixels, one 256x128 / 128x256 tile per CU
    t = f(16384, 512, 18432, 0)
    assert rows[t] * cols[t] == 32768 and blocks(t, 16384, 512) == 256
    # the same layer for a 640-row scan (10 240 pixels): not 320 blocks of 128x128 (two rounds, the second a quarter
    # full) but 640 small ones, three per CU
    t = f(10240, 512, 18432, 0)
    assert rows[t] * cols[t] == 8192 and blocks(t, 10240, 512) == 640
    # bf16 at batch 8: the big layers run on 256x256 tiles
    assert rows[f(131072, 512, 18432, 1)] * cols[f(131072, 512, 18432, 1)] == 65536
    # the stem (Cout 64) only has 64-column tiles to choose from
    assert cols[f(262144, 64, 147, 0)] == 64 and cols[f(262144, 64, 147, 1)] == 64
    # f16x2 (round 4: the model knows how many blocks of a tile a CU holds at once).  Long-K layers: the 128x256 tile of
    # 64x64 wave tiles whose halves run out of phase (5), or two 128x128 blocks per CU (17); a layer with exactly one
    # 128x128 tile per CU and only 256 output channels (layer3 at batch 1) cannot pair anything up and takes the
    # loader-wave tile (14); the short-K 1x1 + identity layers with many blocks per CU take 17; a 640-row scan's head conv
    # takes small tiles, three per CU, like in f32
    assert f(16384, 512, 18432, 2) in (5, 17) and f(16384, 512, 4608, 2) in (5, 17)
    assert f(16384, 256, 2304, 2) == 14 and f(16384, 256, 1024, 2) == 14
    assert f(32768, 256, 2304, 2) in (5, 17)                       # the same layer at batch 2
    assert f(16384, 2048, 512, 2) == 17 and f(16384, 1024, 256, 2) == 17 and f(65536, 256, 64, 2) == 17
    t = f(10240, 512, 18432, 2)
    assert rows[t] * cols[t] == 8192 and blocks(t, 10240, 512) == 640


def test_f16x2_row_normalisation_edge_rows(built_lib, sd_np):
    """Rows the normalisation cannot scale: an all-zero output channel keeps its BatchNorm scale (and packs to zeros); a
    row holding NaN or infinity is normalised by its largest FINITE weight and keeps the NaN / infinity in its pieces (the
    logits then say so, like in the reference); a row of 1e-30 weights hits the clamp of the exponent and still packs to
    finite pieces with a finite scale."""
    name, bn = "backbone.layer2.1.conv1", "backbone.layer2.1.bn1"
    w = sd_np[name + ".weight"].copy()
    w[0] = 0.0
    w[1, 0, 0, 0] = np.nan
    w[2, 1, 0, 0] = np.inf
    w[3] *= np.float32(1e-30)
    sd = dict(sd_np)
    sd[name + ".weight"] = w
    blob, ref = pack_state_dict(sd, "f16x2"), pack_state_dict(sd_np, "f16x2")
    off = 0
    a = lambda v: (v + 255) // 256 * 256
    for u in topology.conv_units():
        ksteps = 7 if u.cin == 3 else u.k * u.k * u.cin * 4 // 128
        wbytes = u.cout * ksteps * 128
        s_off = a(off + wbytes)
        if u.name == name:
            row = ksteps * 128
            scale = blob[s_off: s_off + u.cout * 4].view(np.float32)
            pieces = blob[off: off + wbytes].view(np.float16).reshape(u.cout, -1).astype(np.float32)
            inv = np.float32(1.0) / np.sqrt(sd[bn + ".running_var"] + np.float32(1e-5), dtype=np.float32)
            assert (pieces[0] == 0).all() and scale[0] == sd[bn + ".weight"][0] * inv[0]          # k = 0: the plain BatchNorm scale
            assert np.isnan(pieces[1]).any() and np.isfinite(scale[1]) and np.nanmax(np.abs(pieces[1])) < 2.0 ** 15
            assert np.isinf(pieces[2]).any() and np.isfinite(scale[2])
            assert np.isfinite(pieces[3]).all() and np.isfinite(scale[3]) and scale[3] != 0
            np.testing.assert_array_equal(blob[off + 4 * row: off + wbytes], ref[off + 4 * row: off + wbytes])   # the other rows: untouched
            return
        off = a(a(s_off + u.cout * 4) + u.cout * 4)
    raise AssertionError("unit not found")


def _trailer(blob):
    return blob[-1024:].view(np.int32)


def _unit_index(name):
    return [u.name for u in topology.conv_units()].index(name)


@pytest.mark.parametrize("where,log2_scale", [("internal", -16), ("stream", -20), ("all", -20), ("all", 12)])
def test_f16x2_activation_powers_of_two(built_lib, sd_np, where, log2_scale):
    """nbc_pack_weights stores a tensor whose BatchNorm estimate (max |beta| + 3 |gamma| sqrt(var / (var + eps))) lies outside [2^-5, 2^7] times the
    power of two that brings the estimate to [2, 4): the powers land in the blob's trailer, the producing unit's (scale,
    shift) carry 2^a_out, every reading unit's scale 2^-a_in, classifier.4's f32 weights the last tensor's -- all exact, so
    the rescaled checkpoint packs to the SAME conv weights and to (scale, shift) pairs that are power-of-two multiples of
    the ordinary checkpoint's up to the one rounding rescale_activations itself puts into gamma.  Other precisions: no powers."""
    from conftest import rescale_activations
    s = 2.0 ** log2_scale
    sd = rescale_activations(sd_np, s, where)
    base = pack_state_dict(sd_np, "f16x2")
    blob = pack_state_dict(sd, "f16x2")
    assert built_lib.nbc_packed_weights_flags(blob.ctypes.data, blob.nbytes, 2) == 0
    exps = _trailer(blob)[8:8 + len(topology.conv_units())]
    assert not _trailer(pack_state_dict(sd, "fp32"))[8:].any() and not _trailer(pack_state_dict(sd, "bf16"))[8:].any()
    units = topology.conv_units()
    est = {}
    for i, u in enumerate(units):
        if u.bn is not None:
            var = sd[u.bn + ".running_var"].astype(np.float64)
            est[u.name] = float((np.abs(sd[u.bn + ".bias"]) + 3 * np.abs(sd[u.bn + ".weight"]) * np.sqrt(var / (var + 1e-5))).max())

    def power(e):
        return 0 if 2.0 ** -5 <= e <= 2.0 ** 7 else 1 - int(np.floor(np.log2(e)))

    moved = 0
    for i, u in enumerate(units):
        if u.bn is None:
            assert exps[i] == 0
            continue
        if u.name.endswith(".conv3") or u.name.endswith(".downsample.0"):       # one power per stage: the residual stream's
            stage = u.name[: len("backbone.layerN")]
            e = max(v for k, v in est.items() if k.startswith(stage) and (k.endswith(".conv3") or k.endswith(".downsample.0")))
        else:
            e = est[u.name]
        assert exps[i] == power(e), (u.name, exps[i], e)
        if exps[i] != 0:
            assert 2.0 <= e * 2.0 ** int(exps[i]) < 4.0
            moved += 1
    if where == "internal":
        assert moved == 32 and all(exps[_unit_index(n)] == 0 for n in ("backbone.conv1", "backbone.layer2.1.conv3", "classifier.0"))
    if where == "stream":
        assert moved == 20 and exps[_unit_index("backbone.layer2.1.conv1")] == 0
    if where == "all":
        assert moved == 53 and exps[_unit_index("classifier.0")] == 0           # the head's hidden tensor was not rescaled
    # the conv weight panels do not move at all (their rows are normalised), and what the powers do to (scale, shift) of a
    # producing + reading pair cancels the rescaling: compare with the ordinary checkpoint's blob section by section
    off = 0
    for i, u in enumerate(units):
        if u.bn is None:
            break
        ksteps = 7 if u.cin == 3 else u.k * u.k * u.cin * 4 // 128
        n = u.cout * ksteps * 128
        np.testing.assert_array_equal(blob[off: off + n], base[off: off + n], err_msg=u.name)
        off = (off + n + 255) // 256 * 256
        sc, sc0 = blob[off: off + 4 * u.cout].view(np.float32), base[off: off + 4 * u.cout].view(np.float32)
        off = (off + 4 * u.cout + 255) // 256 * 256
        sh, sh0 = blob[off: off + 4 * u.cout].view(np.float32), base[off: off + 4 * u.cout].view(np.float32)
        off = (off + 4 * u.cout + 255) // 256 * 256
        # stored tensor = 2^a x (s or 1) x the ordinary one; |s 2^a| is whatever put the estimate into [2, 4): at most 2^3 here
        want = sh0.astype(np.float64) * 2.0 ** int(exps[i]) * (s if exps[i] != 0 else 1.0)
        np.testing.assert_allclose(sh, want, rtol=1e-5, atol=1e-6 * np.abs(want).max(), err_msg=u.name)
        ratio = sc.astype(np.float64) / sc0
        assert np.allclose(ratio, ratio.flat[0], rtol=1e-6), u.name             # one factor per unit
        l2 = np.log2(float(ratio.flat[0]))
        assert abs(l2 - round(l2)) < 1e-5, (u.name, ratio.flat[0])


def test_f16x2_activation_powers_keep_an_ordinary_checkpoint_bit_for_bit(built_lib):
    """Every tensor of an ordinary checkpoint (seeds of the trained-like generator, torchvision's plain initialisation) has its
    estimate inside [2^-5, 2^7]: no power, so the blob -- and every result -- is what it was before the powers existed."""
    for kind, seed in (("trained_like", 7), ("trained_like", 11), ("random_init", 5)):
        sd = synth.make_state_dict(kind, seed=seed)
        t = _trailer(pack_state_dict(sd, "f16x2"))
        assert t[1] == 0 and not t[8:].any(), (kind, seed)


def test_pack_flags_report_what_the_normalisations_cannot_reach(built_lib, sd_np):
    """a weight row beyond the row normalisation's clamp (largest |w| below 2^-51 or above 2^81) keeps fewer bits
    than f32 with finite logits: reported (NBC_PACK_ROW_CLAMPED), so that --precision auto goes straight to fp32; a BatchNorm
    scale pushed out of f32's normal range by the powers folded into it likewise (NBC_PACK_SCALE_RANGE).  The accuracy of a
    row INSIDE the clamp's reach is pinned: every weight back to 2^-22 of the row's largest."""
    name, bn = "backbone.layer2.1.conv2", "backbone.layer2.1.bn2"
    for log2_scale, want in ((-45, 0), (-70, 1), (70, 0), (90, 1)):
        sd = dict(sd_np)
        sd[name + ".weight"] = sd_np[name + ".weight"] * np.float32(2.0 ** log2_scale)
        # gamma compensates so that the BatchNorm scale stays an ordinary f32 even for the extreme rows
        blob = pack_state_dict(sd, "f16x2")
        flags = built_lib.nbc_packed_weights_flags(blob.ctypes.data, blob.nbytes, 2)
        assert flags & 1 == want, (log2_scale, flags)
        assert built_lib.nbc_packed_weights_flags(pack_state_dict(sd, "fp32").ctypes.data, blob.nbytes, 0) == 0
        if want == 0:
            # accuracy inside the reach: (P + Q) 2^-k against the f32 weights of channel 0's row
            units = topology.conv_units()
            off = 0
            for u in units:
                ksteps = 7 if u.cin == 3 else u.k * u.k * u.cin * 4 // 128
                if u.name == name:
                    break
                off = (off + u.cout * ksteps * 128 + 255) // 256 * 256
                off += 2 * ((4 * u.cout + 255) // 256 * 256)
            raw = blob[off: off + ksteps * 128]
            g0, g1 = _join_f16x2(raw, ksteps * 32, 32)
            w = sd[name + ".weight"].transpose(0, 2, 3, 1).reshape(u.cout, -1)[0].astype(np.float64)
            k = _row_pow2(w[None].astype(np.float32))[0]
            back = (g0[0].astype(np.float64) + g1[0].astype(np.float64)) / float(k)
            assert np.abs(back - w).max() <= 2.0 ** -22 * np.abs(w).max(), log2_scale
    # a BatchNorm whose scale cannot take its power of two: gamma / sqrt(var + eps) near f32's smallest normal, and a tensor
    # estimate that asks for 2^-a on top
    # a BatchNorm scale that cannot take its powers of two: a tensor at 2^-100 of the usual size (its power: 2^99) read by a
    # convolution whose weights sit at 2^-40 (row power 2^59) under an ordinary BatchNorm: scale x 2^-158 is no f32 normal
    sd = dict(sd_np)
    sd["backbone.layer3.1.bn1.weight"] = sd_np["backbone.layer3.1.bn1.weight"] * np.float32(2.0 ** -100)
    sd["backbone.layer3.1.bn1.bias"] = sd_np["backbone.layer3.1.bn1.bias"] * np.float32(2.0 ** -100)
    sd["backbone.layer3.1.conv2.weight"] = sd_np["backbone.layer3.1.conv2.weight"] * np.float32(2.0 ** -40)
    blob = pack_state_dict(sd, "f16x2")
    assert built_lib.nbc_packed_weights_flags(blob.ctypes.data, blob.nbytes, 2) == 2
    assert built_lib.nbc_packed_weights_flags(pack_state_dict(sd, "fp32").ctypes.data, blob.nbytes, 0) == 0
