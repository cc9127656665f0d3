"""BASELINE.json configs[2] (batch 8, bf16) under test at the full frame size, and the float64
adjudication of the f32 parity mode's label flips.

Tolerances (written here, used below):
  * bf16 throughput mode, batch 8 at 1024x1024: logits within LOGIT_RTOL_BF16 of the oracle's logit range per
    frame, label agreement >= 98 % per frame, every disagreement at an oracle top-2 margin below
    BF16_MARGIN_BAND of the logit range;
  * batch 8 == eight batch-1 calls, bit for bit (labels, counts, low-res logits), both modes;
  * f32 parity mode against the SAME network evaluated in float64 on the CPU (the adjudicator): logits within
    LOGIT_RTOL_FP32 of the logit range; labels IDENTICAL to the float64 labels wherever the float64 top-2
    margin exceeds twice the measured f32 logit error; wherever the HIP f32 labels differ from the CPU f32
    oracle's, the float64 result says which side is right, and the tally is reported (and written to
    gpurun_out/fp64_adjudication_<mode>.json when that directory exists), for both f32-grade modes (fp32, f16x2).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rescale_activations, rescale_conv_weights
from neuralbarkcalculator_amd import synth
from neuralbarkcalculator_amd.model import FCNResNet50

pytestmark = pytest.mark.gpu

LOGIT_RTOL_FP32 = 5e-6      # against float64: measured 1.5e-6 .. 2.3e-6 of the logit range (CPU f32 oracle: 1.2e-6 .. 1.8e-6)
LOGIT_RTOL_BF16 = 4e-2
BF16_MARGIN_BAND = 0.10
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["fp32", "f16x2"])
def gpu_fp32(request, built_lib, sd_np):
    """Both f32-grade modes under the same tolerances (see tests/test_gpu_parity.py)."""
    return FCNResNet50(request.param).load_state_dict(sd_np).to(DEV)


@pytest.fixture(scope="module")
def gpu_bf16(built_lib, sd_np):
    return FCNResNet50("bf16").load_state_dict(sd_np).to(DEV)


@pytest.fixture(scope="module")
def oracle_f64(sd_np):
    """The oracle's topology with every parameter and activation in float64 (torch CPU): the adjudicator."""
    from oracle.fcn_resnet50_oracle import OracleFCNResNet50
    m = OracleFCNResNet50()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    return m.double()


def frames(idx, h, w):
    return torch.from_numpy(np.stack([synth.make_input(int(i), h, w) for i in idx]))


def test_config2_batch8_bf16_at_1024_vs_oracle(oracle_model, gpu_bf16):
    """configs[2]: ONE batch of 8 distinct 1024x1024 frames through the bf16 path vs the oracle, frame by frame."""
    from oracle.fcn_resnet50_oracle import predict_labels
    idx = list(range(50, 58))
    x = frames(idx, 1024, 1024)
    xd = x.to(DEV)
    gpu_bf16.autotune(xd, objective="latency")             # the tiles bench.py's configs[2] leg runs with
    labels, counts, lowres = gpu_bf16.predict_labels(xd, labels_dtype=torch.uint8, return_lowres=True)
    logits = gpu_bf16(xd)
    torch.cuda.synchronize()
    assert tuple(labels.shape) == (8, 1024, 1024) and tuple(counts.shape) == (8, 3)
    assert torch.equal(labels.long(), torch.argmax(logits, 1))
    assert counts.sum(1).tolist() == [1024 * 1024] * 8
    worst_agree, worst_err = 1.0, 0.0
    for b, i in enumerate(idx):
        labels_ref, counts_ref, logits_ref, lowres_ref = predict_labels(oracle_model, x[b:b + 1])
        scale = float(logits_ref.abs().max())
        err = float((logits[b].cpu() - logits_ref[0]).abs().max())
        assert err <= LOGIT_RTOL_BF16 * scale, (i, err, scale)
        mism = labels[b].cpu().long() != labels_ref[0]
        agree = 1.0 - float(mism.float().mean())
        assert agree >= 0.98, (i, agree)
        if mism.any():
            top2 = torch.topk(logits_ref, 2, dim=1).values
            margin = (top2[:, 0] - top2[:, 1])[0]
            assert float(margin[mism].max()) <= BF16_MARGIN_BAND * scale, (i, float(margin[mism].max()), scale)
        assert (counts_ref[0] > 0.02 * 1024 * 1024).all(), counts_ref       # a frame with all three classes
        worst_agree, worst_err = min(worst_agree, agree), max(worst_err, err / scale)
    print("configs[2] batch 8 bf16 @1024: worst label agreement %.5f, worst rel. logit error %.4f" % (worst_agree, worst_err))


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_batch8_equals_eight_singles_at_1024(gpu_fp32, gpu_bf16, mode):
    """A frame's result does not depend on the batch it rides in: batch 8 == 8 x batch 1, bit for bit."""
    model = gpu_fp32 if mode == "fp32" else gpu_bf16
    x = frames(range(60, 68), 1024, 1024).to(DEV)
    labels, counts, lowres = model.predict_labels(x, labels_dtype=torch.uint8, return_lowres=True)
    torch.cuda.synchronize()
    for b in range(8):
        l1, c1, r1 = model.predict_labels(x[b:b + 1], labels_dtype=torch.uint8, return_lowres=True)
        assert torch.equal(r1[0], lowres[b]), f"frame {b}: low-res logits differ ({mode})"
        assert torch.equal(l1[0], labels[b]) and torch.equal(c1[0], counts[b]), f"frame {b} ({mode})"


@pytest.mark.parametrize("name", ["c128", "b2_256", "full1024"])
def test_fp32_label_flips_adjudicated_by_float64(oracle_model, oracle_f64, gpu_fp32, name):
    from oracle.fcn_resnet50_oracle import predict_labels
    g = load_golden(name)
    h, w = (int(v) for v in g["hw"])
    x = frames(g["frames"], h, w)
    lab32, _, log32, _ = predict_labels(oracle_model, x)                 # CPU f32 oracle
    lab64, _, log64, _ = predict_labels(oracle_f64, x.double())          # float64 adjudicator
    xd = x.to(DEV)
    labels, counts = gpu_fp32.predict_labels(xd)
    logits = gpu_fp32(xd)
    torch.cuda.synchronize()
    labels, logits = labels.cpu(), logits.cpu().double()
    scale = float(log64.abs().max())
    err_hip = float((logits - log64).abs().max())
    err_cpu = float((log32.double() - log64).abs().max())
    assert err_hip <= LOGIT_RTOL_FP32 * scale, (err_hip, scale)
    top2 = torch.topk(log64, 2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    # (i) outside the tie band the integer labels are the float64 labels
    clear = margin > 2.0 * err_hip
    assert torch.equal(labels[clear], lab64[clear]), "an HIP f32 label differs from float64 outside the tie band"
    # (ii) every pixel where HIP f32 and the CPU f32 oracle disagree: who agrees with float64?
    flip = labels != lab32
    n_flip = int(flip.sum())
    hip_right = int((labels[flip] == lab64[flip]).sum())
    cpu_right = int((lab32[flip] == lab64[flip]).sum())
    report = {"case": name, "mode": gpu_fp32.precision, "pixels": int(labels.numel()), "logit_range": scale,
              "max_abs_logit_err_hip_f32_vs_f64": err_hip, "max_abs_logit_err_cpu_f32_vs_f64": err_cpu,
              "labels_hip_f32_vs_cpu_f32_oracle_differ": n_flip,
              "of_those_hip_agrees_with_f64": hip_right, "of_those_cpu_oracle_agrees_with_f64": cpu_right,
              "labels_hip_f32_vs_f64_differ": int((labels != lab64).sum()),
              "labels_cpu_f32_vs_f64_differ": int((lab32 != lab64).sum()),
              "max_f64_margin_at_any_hip_vs_f64_difference": float(margin[labels != lab64].max()) if bool((labels != lab64).any()) else 0.0,
              "pixels_inside_tie_band_2x_err_hip": int((~clear).sum())}
    print(json.dumps(report))
    # a flipped pixel is a tie: with three classes one of the two sides carries the float64 label
    assert hip_right + cpu_right >= n_flip
    assert n_flip <= int((~clear).sum())
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        path = os.path.join(out, "fp64_adjudication_%s.json" % gpu_fp32.precision)
        prev = json.load(open(path)) if os.path.exists(path) else {}
        prev[name] = report
        json.dump(prev, open(path, "w"), indent=1)


@pytest.mark.parametrize("kind,seed,scale", [("trained_like", 11, 1.0), ("trained_like", 23, 1.0), ("random_init", 5, 1.0),
                                             ("trained_like", 7, 64.0), ("trained_like", 7, 1.0 / 64.0)])
def test_f32_grade_modes_on_other_weights(built_lib, kind, seed, scale):
    """The f32 tolerances are not a property of one weight set: other seeds, torchvision's plain initialisation (whose
    activations grow through the residual stream: BN gamma 1 everywhere), and the seed-7 network with its stem scaled up
    and down by 64 (activations of the first stages 64 times larger / smaller: the low f16 pieces of small values become
    subnormal, large values approach f16's range) -- f16x2 and the f32 MFMA mode against the CPU oracle AND a float64
    evaluation, logits within LOGIT_RTOL_FP32 of the range; labels equal to float64's outside the tie band."""
    from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels
    sd = synth.make_state_dict(kind, seed=seed)
    if scale != 1.0:
        sd["backbone.conv1.weight"] = sd["backbone.conv1.weight"] * np.float32(scale)
        # the next BatchNorm would undo a scale in front of it: scale its statistics along, so that the stem's OUTPUT
        # (after bn1 + ReLU) really is `scale` times larger and everything behind sees other magnitudes
        sd["backbone.bn1.weight"] = sd["backbone.bn1.weight"] * np.float32(scale)
        sd["backbone.bn1.running_mean"] = sd["backbone.bn1.running_mean"] * np.float32(scale)
        sd["backbone.bn1.running_var"] = sd["backbone.bn1.running_var"] * np.float32(scale * scale)
    oracle = OracleFCNResNet50()
    oracle.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    o64 = OracleFCNResNet50()
    o64.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    o64 = o64.double()
    x = frames([81, 82], 256, 320)
    lab32, _, log32, low32 = predict_labels(oracle, x)
    lab64, _, log64, low64 = predict_labels(o64, x.double())
    rng = float(log64.abs().max())
    report = {}
    for mode in ("f16x2", "fp32"):
        m = FCNResNet50(mode).load_state_dict(sd).to(DEV)
        labels, counts, lowres = m.predict_labels(x.to(DEV), return_lowres=True)
        logits = m(x.to(DEV))
        torch.cuda.synchronize()
        assert not m.nonfinite_seen()
        err64 = float((logits.cpu().double() - log64).abs().max())
        err32 = float((lowres.cpu() - low32).abs().max())
        report[mode] = (err64 / rng, err32 / rng)
        assert err64 <= LOGIT_RTOL_FP32 * rng and err32 <= LOGIT_RTOL_FP32 * rng, (mode, kind, seed, scale, err64, err32, rng)
        top2 = torch.topk(log64, 2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2.0 * err64
        assert torch.equal(labels.cpu()[clear], lab64[clear]), (mode, kind, seed, scale)
    cpu_err = float((log32.double() - log64).abs().max()) / rng
    print("%s seed %d stem x%g: logit error / range vs float64: f16x2 %.2e, f32 MFMA %.2e, CPU f32 oracle %.2e"
          % (kind, seed, scale, report["f16x2"][0], report["fp32"][0], cpu_err))


@pytest.mark.parametrize("log2_scale", [-12, -16, 10])
def test_f32_grade_modes_with_rescaled_conv_weights(built_lib, sd_np, log2_scale):
    """The small-magnitude floor of the f16 pieces (what's weak 2): with every convolution's weights at
    2^-12 or 2^-16 of their usual magnitude (|w| around 5e-6 and 3e-7) the low piece of a weight, split as it stands, is
    an f16 subnormal or zero and the product loses up to half its bits -- with finite logits, so nothing raises the
    non-finite flag.  nbc_pack_weights therefore normalises every output channel's weight row by a power of two before
    the split and folds the inverse into the f32 BatchNorm scale (exact).  f16x2 and the f32 MFMA mode against the CPU
    f32 oracle AND a float64 evaluation of the same state_dict under LOGIT_RTOL_FP32; labels equal to float64's outside
    the tie band.  2^10 checks the other direction (weights of 20 .. 100)."""
    from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels
    sd = rescale_conv_weights(sd_np, 2.0 ** log2_scale)
    oracle = OracleFCNResNet50()
    oracle.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    o64 = OracleFCNResNet50()
    o64.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    o64 = o64.double()
    x = frames([81, 82], 256, 320)
    lab32, counts32, log32, low32 = predict_labels(oracle, x)
    lab64, _, log64, low64 = predict_labels(o64, x.double())
    assert (counts32 > 0.02 * 256 * 320).all(), counts32          # still a frame with all three classes
    rng = float(log64.abs().max())
    report, failures = {}, []
    for mode in ("f16x2", "fp32"):
        m = FCNResNet50(mode).load_state_dict(sd).to(DEV)
        labels, counts, lowres = m.predict_labels(x.to(DEV), return_lowres=True)
        logits = m(x.to(DEV))
        torch.cuda.synchronize()
        assert not m.nonfinite_seen()
        err64 = float((logits.cpu().double() - log64).abs().max())
        err32 = float((lowres.cpu() - low32).abs().max())
        report[mode] = (err64 / rng, err32 / rng)
        top2 = torch.topk(log64, 2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2.0 * err64
        if not (err64 <= LOGIT_RTOL_FP32 * rng and err32 <= LOGIT_RTOL_FP32 * rng):
            failures.append((mode, "logits", err64 / rng, err32 / rng))
        if not torch.equal(labels.cpu()[clear], lab64[clear]):
            failures.append((mode, "labels outside the tie band"))
    cpu_err = float((log32.double() - log64).abs().max()) / rng
    print("conv weights x 2^%d: logit error / range vs float64: f16x2 %.2e, f32 MFMA %.2e, CPU f32 oracle %.2e; vs the CPU oracle "
          "(low-res): f16x2 %.2e, f32 MFMA %.2e" % (log2_scale, report["f16x2"][0], report["fp32"][0], cpu_err,
                                                    report["f16x2"][1], report["fp32"][1]))
    assert not failures, failures


@pytest.mark.parametrize("where,log2_scale", [("internal", -16), ("internal", -20), ("stream", -16), ("stream", -20),
                                              ("all", -20), ("all", 12)])
def test_f32_grade_modes_with_rescaled_activations(built_lib, sd_np, where, log2_scale):
    """The activation side of the f16 pieces' floor (what's weak 2 / item 4).  A tensor between a BatchNorm and
    the next convolution is scale-free, so a checkpoint may hold bottleneck activations (or a residual stream) at 2^-16 or
    2^-20 of their usual size; split as they stand, the LOW piece of such a value is an f16 subnormal or zero and every
    product loses bits -- with finite logits, so nothing raises the non-finite flag (measured on round 4's library:
    profiles/r05_small_activation_floor_before_fix.log).  At 2^12 the other end: values beyond 65 504 overflow the high
    piece.  nbc_pack_weights therefore estimates every tensor's magnitude from its producing BatchNorm (max |beta| +
    3 |gamma| sqrt(var / (var + eps))), picks a power of two per tensor (one per residual stream) and folds it into the producer's f32 (scale,
    shift) and its inverse into every consumer's scale: exact, ReLU and max-pool being positively homogeneous.
    f16x2 and the f32 MFMA mode against the CPU f32 oracle AND a float64 evaluation of the same state_dict under
    LOGIT_RTOL_FP32; labels equal to float64's outside the tie band; layer-by-layer read-back takes the power off again."""
    from oracle.fcn_resnet50_oracle import OracleFCNResNet50, predict_labels
    sd = rescale_activations(sd_np, 2.0 ** log2_scale, where)
    oracle = OracleFCNResNet50()
    oracle.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    o64 = OracleFCNResNet50()
    o64.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    o64 = o64.double()
    x = frames([81, 82], 256, 320)
    lab32, counts32, log32, low32 = predict_labels(oracle, x)
    lab64, _, log64, low64 = predict_labels(o64, x.double())
    assert (counts32 > 0.02 * 256 * 320).all(), counts32          # still a frame with all three classes
    rng = float(log64.abs().max())
    report, failures = {}, []
    for mode in ("f16x2", "fp32"):
        m = FCNResNet50(mode).load_state_dict(sd).to(DEV)
        labels, counts, lowres = m.predict_labels(x.to(DEV), return_lowres=True)
        logits = m(x.to(DEV))
        torch.cuda.synchronize()
        if m.nonfinite_seen():
            failures.append((mode, "non-finite logits"))
            report[mode] = (float("nan"), float("nan"))
            continue
        err64 = float((logits.cpu().double() - log64).abs().max())
        err32 = float((lowres.cpu() - low32).abs().max())
        report[mode] = (err64 / rng, err32 / rng)
        top2 = torch.topk(log64, 2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 2.0 * err64
        if not (err64 <= LOGIT_RTOL_FP32 * rng and err32 <= LOGIT_RTOL_FP32 * rng):
            failures.append((mode, "logits", err64 / rng, err32 / rng))
        if not torch.equal(labels.cpu()[clear], lab64[clear]):
            failures.append((mode, "labels outside the tie band"))
    cpu_err = float((log32.double() - log64).abs().max()) / rng
    print("activations (%s) x 2^%d: logit error / range vs float64: f16x2 %.2e, f32 MFMA %.2e, CPU f32 oracle %.2e; vs the CPU "
          "oracle (low-res): f16x2 %.2e, f32 MFMA %.2e" % (where, log2_scale, report["f16x2"][0], report["fp32"][0], cpu_err,
                                                           report["f16x2"][1], report["fp32"][1]))
    assert not failures, failures
