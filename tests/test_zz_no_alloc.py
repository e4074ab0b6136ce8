"""fp_attn_fwd_bf16_static (the entry point that allocates nothing: its split-round partials live in a fixed device
array inside the library) against fp_attn_fwd_bf16 with a caller workspace, through the C ABI.  Same kernels, same
split, same merge order: the outputs must be bit-identical.  Two calls in a row on one stream, the second on a smaller
input (the array keeps the first call's partials), then a call that does not split and a varlen call.

"Allocates nothing" is checked as far as a test outside the library can: after one warm-up call (the first call on a
device looks up the array's address and loads the kernels' code object), the device's free memory (`hipMemGetInfo`
through torch.cuda.mem_get_info: it sees every hipMalloc of the process, the library's included, which torch's own
counters would not) is the same before and after the timed calls, with torch's allocator quiescent in between.

Collected last on purpose (the round-end run is `pytest -x`)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from framepack_amd import lib as fplib  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def _call(fn_name, q, k, v, heads, cu=None, max_seqlen=None, ws=None, out=None):
    L = fplib.load()
    S = q.shape[0]
    o = torch.full_like(q, float("nan")) if out is None else out
    nseg = 1 if cu is None else cu.numel() - 1
    args = [q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), q.stride(0), k.stride(0), v.stride(0), o.stride(0),
            None if cu is None else cu.data_ptr(), nseg, S if max_seqlen is None else max_seqlen, S, heads, 128,
            1.0 / math.sqrt(128.0)]
    if fn_name == "fp_attn_fwd_bf16":
        args += [ws.data_ptr(), ws.numel()]
    args.append(torch.cuda.current_stream().cuda_stream)
    fplib.check(getattr(L, fn_name)(*args), fn_name)
    return o


def _qkv(S, heads, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return tuple(torch.randn(S, heads * 128, generator=g, device=DEV).to(BF) for _ in range(3))


def test_static_scratch_entry_point_matches_workspace_entry_point():
    L = fplib.load()
    # the array holds one partial slot per CU of an MI355X (a caller's workspace of fp_attn_workspace_bytes() two): on a
    # GPU with more CUs than slots the limit would bite.  The workspace call below gets a workspace of the array's size:
    # the split follows the slot count, and the same split is what makes the two outputs the same bits.
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    assert L.fp_attn_static_scratch_bytes() >= 256 + ncu * (256 * 128 + 512) * 4
    ws = torch.empty((L.fp_attn_static_scratch_bytes(),), dtype=torch.uint8, device=DEV)
    # 260 and 261 work items on 256 CUs: the last round's 4 / 5 items are swept as 8 KV ranges each and merged
    cases = [(130 * 256 - 100, 2, None, None), (86 * 256 + 10, 3, None, None), (300, 2, None, None),
             (1064, 2, [0, 300, 517, 517, 1000, 1064], 483)]
    # inputs and outputs first, then a warm-up call: from here to `after` nothing on the Python side allocates
    prepared = []
    for i, (S, heads, cu, mx) in enumerate(cases):
        q, k, v = _qkv(S, heads, seed=10 + i)
        cu_t = None if cu is None else torch.tensor(cu, dtype=torch.int32, device=DEV)
        prepared.append((q, k, v, heads, cu_t, mx, torch.full_like(q, float("nan"))))
    for q, k, v, heads, cu_t, mx, o in prepared:
        _call("fp_attn_fwd_bf16_static", q, k, v, heads, cu_t, mx, out=o)
        o.fill_(float("nan"))
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    reserved_before = torch.cuda.memory_reserved()
    outs = []
    for q, k, v, heads, cu_t, mx, o in prepared:
        got = _call("fp_attn_fwd_bf16_static", q, k, v, heads, cu_t, mx, out=o)
        outs.append((got, q, k, v, heads, cu_t, mx))
    torch.cuda.synchronize()
    assert torch.cuda.memory_reserved() == reserved_before          # torch's allocator took nothing from the device ...
    assert torch.cuda.mem_get_info()[0] == free_before               # ... and neither did the library (no hipMalloc)
    for got, q, k, v, heads, cu_t, mx in outs:
        ref = _call("fp_attn_fwd_bf16", q, k, v, heads, cu_t, mx, ws=ws)
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got, ref)
    # against the fp32 math too, on rows of a split item of the second (smaller) call
    got, q, k, v, heads = outs[1][:5]
    sl = slice(2 * 128, 3 * 128)
    lo, hi = 82 * 256, q.shape[0]
    p = torch.softmax(q[lo:hi, sl].float() @ k[:, sl].float().T / math.sqrt(128.0), dim=-1)
    ref32 = p @ v[:, sl].float()
    err = ((got[lo:hi, sl].float() - ref32).norm() / ref32.norm()).item()
    assert err < 5e-3
    del outs


def test_static_scratch_entry_point_rejects_bad_arguments():
    q, k, v = _qkv(64, 1, seed=1)
    L = fplib.load()
    err = L.fp_attn_fwd_bf16_static(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.data_ptr(), 128, 128, 128, 128, None, 1,
                                    64, 64, 1, 64, 0.1, torch.cuda.current_stream().cuda_stream)   # head_dim 64
    assert err == 1
