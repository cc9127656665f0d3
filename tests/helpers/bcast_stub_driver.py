"""Child process of test_bcast_weights_non_root_branch_with_a_stub: argv[1] = the built rccl stub.
Loads the stub RTLD_GLOBAL first, so that libnbc_hip.so's dlsym(RTLD_DEFAULT, "ncclBroadcast") finds it (torch's own
librccl is a private dependency of libtorch_hip and never enters the global scope), then plays rank 1 of 2."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
stub = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)
stub.stub_plant_root_blob.argtypes = [C.c_void_p, C.c_size_t]
stub.stub_last_count.restype = C.c_size_t

import numpy as np   # noqa: E402
import torch         # noqa: E402
from neuralbarkcalculator_amd import _lib, synth   # noqa: E402
from neuralbarkcalculator_amd.model import FCNResNet50   # noqa: E402

DEV = torch.device("cuda", 0)
lib = _lib.load()
x = torch.from_numpy(np.stack([synth.make_input(74, 64, 64), synth.make_input(75, 64, 64)])).to(DEV)
stream = torch.cuda.current_stream(DEV).cuda_stream
comm = C.c_void_p(0x1234)                      # opaque to the library; the stub never dereferences it


def lowres_through_abi(ctx):
    out = torch.empty((2, 3, 8, 8), dtype=torch.float32, device=DEV)
    _lib.check(lib.nbc_forward(ctx, x.data_ptr(), _lib.IN_F32_NCHW, 2, 64, 64, out.data_ptr(), None, None, _lib.LABEL_U8,
                               None, 0, stream), "nbc_forward")
    torch.cuda.synchronize()
    return out


for prec_name, prec in (("bf16", _lib.PREC_BF16), ("fp32", _lib.PREC_FP32)):
    sd = synth.make_state_dict("trained_like", seed=7)
    root = FCNResNet50(prec_name).load_state_dict(sd).to(DEV)          # "rank 0": read the checkpoint
    want = root.lowres_logits(x)
    nbytes = lib.nbc_packed_weights_bytes(prec)
    stub.stub_plant_root_blob(C.c_void_p(root._blob_dev.data_ptr()), nbytes)

    recv = FCNResNet50(prec_name).to(DEV)                              # "rank 1": a context without weights
    stub.stub_set_rank(1)
    assert lib.nbc_forward(recv._ctx, x.data_ptr(), _lib.IN_F32_NCHW, 2, 64, 64, None, None, None, _lib.LABEL_U8, None, 0,
                           stream) == _lib.NBC_ERR_STATE          # nothing attached yet
    n0 = stub.stub_calls()
    _lib.check(lib.nbc_bcast_weights(recv._ctx, comm, 0, prec, stream), "nbc_bcast_weights (non-root)")
    assert stub.stub_calls() == n0 + 1 and stub.stub_last_count() == nbytes
    assert torch.equal(lowres_through_abi(recv._ctx), want), "forward on the received blob differs"

    # a second broadcast into a context that already owns a received blob: the old one is freed, the new one attached
    sd2 = {k: (v * np.float32(0.5) if k == "classifier.4.weight" else v) for k, v in sd.items()}
    root2 = FCNResNet50(prec_name).load_state_dict(sd2).to(DEV)
    want2 = root2.lowres_logits(x)
    assert not torch.equal(want2, want)
    stub.stub_plant_root_blob(C.c_void_p(root2._blob_dev.data_ptr()), nbytes)
    _lib.check(lib.nbc_bcast_weights(recv._ctx, comm, 0, prec, stream), "nbc_bcast_weights (second)")
    assert torch.equal(lowres_through_abi(recv._ctx), want2)

    # a failing collective leaves the context on the weights it had
    stub.stub_fail_next(5)
    assert lib.nbc_bcast_weights(recv._ctx, comm, 0, prec, stream) == _lib.NBC_ERR_HIP
    assert b"ncclBroadcast returned 5" in lib.nbc_last_error()
    assert torch.equal(lowres_through_abi(recv._ctx), want2)

    # the root side of the same stub communicator: sends what it holds, keeps it
    stub.stub_set_rank(0)
    _lib.check(lib.nbc_bcast_weights(root._ctx, comm, 0, prec, stream), "nbc_bcast_weights (root)")
    assert torch.equal(root.lowres_logits(x), want)
    # a receiver asked for a precision other than the one broadcast gets that layout (sizes differ per precision)
    print("stub broadcast OK:", prec_name, nbytes, "bytes")
print("bcast stub driver OK")
