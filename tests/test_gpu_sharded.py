"""-m gpu: the view-sharded path on ONE GPU.

(a) the real ViewSharding control flow over RCCL with world_size=1 (HipExecutor, split QKV
    prologue, K/V^T all-gather buffers, segment attention, camera-token gather) must reproduce the
    unsharded forward;
(b) two uneven shards (2+1 views) emulated sequentially in one process -- each "rank" runs the
    HipExecutor steps on its views, the all-gather is a torch.stack -- must reproduce the
    monolithic result (this is the multi-rank numerics: K/V^T padded to the largest shard,
    per-segment valid counts, rank-ordered segments, q-only / kv-only QKV launches).
The 2/4/8-GPU run itself is the driver's; tests/test_sharding_gloo.py covers the multi-process
control flow on CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import common
from omnivggt_official_amd import lib as L, sharding
from omnivggt_official_amd.model import OmniVGGT

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(depth, dino, dtype):
    sd = common.reduced_state_dict(depth, dino)
    with torch.device("meta"):
        m = OmniVGGT(depth=depth, dino_depth=dino, compute_dtype=dtype)
    m = m.to_empty(device="cpu")
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def test_world_size_one_rccl_matches_unsharded():
    L.require_gpu()
    m = build(2, 1, torch.bfloat16)
    S, dgi, cgi = 3, [1], [0, 2]
    inp = common.inputs_for(S, DEV)
    args = (inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
    with torch.no_grad():
        ref, _ = m.aggregator(*args)
        ref_out = m(*args)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m.aggregator.shard = sharding.ViewSharding(gather_output=False)
        with torch.no_grad():
            got, start = m.aggregator(*args)
            out = m(*args)
        assert start == 5 and m.aggregator.shard.last_partition == [(0, 3)]
        for a, b in zip(got, ref):
            assert a.shape == b.shape
            assert common.max_rel(a.cpu(), b.cpu()) <= 1e-6        # same kernels, same order
        for key in ("pose_enc", "depth", "world_points"):
            assert common.max_rel(out[key].cpu(), ref_out[key].cpu()) <= 1e-5
        # the head-parallel (all_to_all_single) exchange through real RCCL, world size 1
        m.aggregator.shard = sharding.ViewSharding(gather_output=False, mode="heads")
        with torch.no_grad():
            got2, _ = m.aggregator(*args)
        for a, b in zip(got2, ref):
            assert common.max_rel(a.cpu(), b.cpu()) <= 1e-6
        # bench.py's self-check that picks the exchange form
        rep = m.aggregator.shard.choose_mode(lambda: m.aggregator(*args)[0][-1], S)
        assert rep["exchange"] == "head-parallel all-to-all" and rep["selfcheck_max_rel_vs_allgather"] <= 1e-6
        assert m.aggregator.shard.mode == "heads"
    finally:
        m.aggregator.shard = None
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_two_uneven_shards_emulated_match_monolithic(dtype, tol):
    L.require_gpu()
    m = build(2, 1, dtype)
    agg = m.aggregator
    S, dgi, cgi = 3, [1], [0, 2]
    inp = common.inputs_for(S, DEV)
    inputs = (inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
    with torch.no_grad():
        ref, _ = agg(*inputs)
    P, C = agg.tokens_per_view, 1024
    parts = sharding.partition(S, 2)
    assert parts == [(0, 2), (2, 3)]
    counts = [(h - l) * P for l, h in parts]
    max_local = max(h - l for l, h in parts)
    ranks = []
    with torch.no_grad():
        for r, (lo, hi) in enumerate(parts):
            ex = sharding.HipExecutor(agg, torch.device(DEV))
            tokens0, tables = ex.embed(inputs, (lo, hi))
            ws_f, ws_g = ex.workspaces(hi - lo, max_local, P)
            ranks.append(dict(ex=ex, x=tokens0, tables=tables, ws_f=ws_f, ws_g=ws_g, outs=ex.new_outputs(hi - lo, P), lo=lo, hi=hi))
        for i in range(agg.depth):
            ks, vs = [], []
            for st in ranks:
                buf = st["outs"][i].view(-1, 2 * C)
                st["ex"].frame_block(i, st["ws_f"], st["x"], buf[:, :C], st["tables"][i + 1][st["lo"]:st["hi"]].contiguous(), P)
                k, vt = st["ex"].global_kv(i, st["ws_g"], buf[:, :C], buf[:, C:])
                ks.append(k.clone())
                vs.append(vt.clone())
            kg, vg = torch.stack(ks), torch.stack(vs)                  # the all-gather
            for r, st in enumerate(ranks):
                buf = st["outs"][i].view(-1, 2 * C)
                st["ex"].global_q(i, st["ws_g"], buf[:, :C], buf[:, C:])
                st["ex"].global_rest(i, st["ws_g"], buf[:, :C], buf[:, C:], kg, vg, counts, r)
                st["x"] = buf[:, C:]
    for i in range(agg.depth):
        got = torch.cat([st["outs"][i] for st in ranks], dim=1)
        assert got.shape == ref[i].shape
        assert common.max_rel(got.cpu(), ref[i].cpu()) <= tol


def test_two_even_shards_head_parallel_emulated_match_monolithic():
    """The all-to-all (head-parallel) exchange, two ranks of 2 views emulated sequentially in one process: per-rank QKV
    for all 16 heads, chunk exchange (= all_to_all_single), attention over (source, head) batch entries with kv_heads = 8
    and two K / V^T segments, head-major outputs exchanged back, head-major -> token-major copy, epilogue without the
    attention launch -- must reproduce the unsharded forward."""
    L.require_gpu()
    m = build(2, 1, torch.bfloat16)
    agg = m.aggregator
    S, dgi, cgi, world = 4, [1, 2], [0, 3], 2
    inp = common.inputs_for(S, DEV)
    inputs = (inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
    with torch.no_grad():
        ref, _ = agg(*inputs)
    P, C = agg.tokens_per_view, 1024
    parts = sharding.partition(S, world)
    hpr = 16 // world
    ranks = []

    def exchange(send):                       # all_to_all_single: rank r receives chunk r of every rank's send buffer
        return [torch.cat([send[s][r * hpr:(r + 1) * hpr] for s in range(world)], dim=0).contiguous() for r in range(world)]

    with torch.no_grad():
        for r, (lo, hi) in enumerate(parts):
            ex = sharding.HipExecutor(agg, torch.device(DEV))
            tokens0, tables = ex.embed(inputs, (lo, hi))
            ws_f, ws_g, xb = ex.heads_workspaces(hi - lo, P)
            ranks.append(dict(ex=ex, x=tokens0, tables=tables, ws_f=ws_f, ws_g=ws_g, xb=xb, outs=ex.new_outputs(hi - lo, P), lo=lo, hi=hi))
        n = (parts[0][1] - parts[0][0]) * P
        for i in range(agg.depth):
            sq, sk, sv = [], [], []
            for st in ranks:
                buf = st["outs"][i].view(-1, 2 * C)
                st["ex"].frame_block(i, st["ws_f"], st["x"], buf[:, :C], st["tables"][i + 1][st["lo"]:st["hi"]].contiguous(), P)
                q, k, vt = st["ex"].global_qkv(i, st["ws_g"], buf[:, :C], buf[:, C:])
                sq.append(q.clone()); sk.append(k.clone()); sv.append(vt.clone())     # the workspaces are shared between the emulated ranks
            rq, rk, rv = exchange(sq), exchange(sk), exchange(sv)
            so = [st["ex"].head_attention(rq[r], rk[r], rv[r], torch.empty_like(rq[r]), n, world) for r, st in enumerate(ranks)]
            ro = exchange(so)
            for r, st in enumerate(ranks):
                buf = st["outs"][i].view(-1, 2 * C)
                st["ex"].global_finish(i, st["ws_g"], buf[:, :C], buf[:, C:], ro[r], n)
                st["x"] = buf[:, C:]
    for i in range(agg.depth):
        got = torch.cat([st["outs"][i] for st in ranks], dim=1)
        assert got.shape == ref[i].shape
        assert common.max_rel(got.cpu(), ref[i].cpu()) <= 2e-2
