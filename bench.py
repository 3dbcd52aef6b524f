#!/usr/bin/env python
"""Benchmark of the OmniVGGT aggregator hot path on MI355X (contract: see the task prompt).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--views S] [--dtype bf16|f16|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE forward of the aggregator (DINOv2 embed + modality fusion + 24 x [frame block,
camera injection, global block]) over one synthetic multi-view batch already resident in HBM.
Default workload at every N: 64 views, 518^2, images-only, bf16 -- BASELINE.json configs[3] and the
north star's scaling statement ("frames/sec at 1/2/4/8 GPUs ... on 64-view 518^2 synthetic input";
SURVEY.md section 8d: "Config 4 ... also run G=1,2,4 for the scaling curve"), so `python bench.py --gpus N`
for N=1,2,4,8 IS that strong-scaling curve ("scaling": "strong": total work fixed, views sharded
over the ranks; global attention exchanged by the head-parallel all-to-all of sharding.py over RCCL, `--shard-mode allgather`
selects the K/V^T all-gather form).  At N=1 the same JSON line also carries `secondary`:
the 8-view configs[1] measurement (frames/s + roofline of the same kernel on that shape).
--views S overrides the view count; --aux adds depth + camera tokens on every view (configs[2]).

Prints ONE JSON line on rank 0 (metric frames/s, roofline of the global-attention kernel
measured live with HIP events inside the timed steps, cpu_baseline = oracle on host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from omnivggt_official_amd import lib as L, weights  # noqa: E402
from omnivggt_official_amd.model import OmniVGGT  # noqa: E402

P_TOK, C = 1374, 1024
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}     # MI355X_MICROARCH.md, dense MFMA
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def agg_flops(S):
    """F(S) and F_ga(S) for B=1 (FLOP)."""
    f_ga = 96.0 * (S * P_TOK) ** 2 * 1024
    return S * (72 * 34.58e9 + 48 * 7.73e9 + 1.65e9) + f_ga, f_ga


def synthetic_inputs(S, device, seed=1234, aux=False):
    """SURVEY.md section 8d inputs: images U[0,1); with aux also random-rotation cameras, depth 0.5+5U,
    mask Bernoulli(0.8).  Images-only runs carry the loader's zero placeholders (visual_util.py:793-824)."""
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(1, S, 3, 518, 518, generator=g)
    z = lambda *s: torch.zeros(*s)
    if not aux:
        return dict(images=images.to(device), extrinsics=z(1, S, 3, 4).to(device), intrinsics=z(1, S, 3, 3).to(device),
                    depth=z(1, S, 518, 518, 1).to(device), mask=z(1, S, 518, 518).to(device))
    from omnivggt_official_amd.camera_math import quaternion_to_rotation
    q = torch.randn(S, 4, generator=g)
    R = quaternion_to_rotation(q / q.norm(dim=-1, keepdim=True))
    ext = torch.cat([R, torch.randn(S, 3, 1, generator=g)], dim=-1).unsqueeze(0)
    f = 400 + 300 * torch.rand(S, generator=g)
    K = torch.zeros(1, S, 3, 3)
    K[0, :, 0, 0], K[0, :, 1, 1], K[0, :, 0, 2], K[0, :, 1, 2], K[0, :, 2, 2] = f, f, 259.0, 259.0, 1.0
    depth = 0.5 + 5 * torch.rand(1, S, 518, 518, 1, generator=g)
    mask = (torch.rand(1, S, 518, 518, generator=g) > 0.2).float()
    return dict(images=images.to(device), extrinsics=ext.to(device), intrinsics=K.to(device), depth=depth.to(device), mask=mask.to(device))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views", type=int, default=0, help="total views S (default 64 at every N)")
    ap.add_argument("--dtype", default="bf16", choices=list(DT))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-heads", action="store_true", help="with --e2e: run the DPT heads as f32 PyTorch modules instead of the HIP kernels")
    ap.add_argument("--e2e-views", type=int, default=8, help="view count of the --e2e measurement")
    ap.add_argument("--e2e", action="store_true", help="also time OmniVGGT.forward incl. the PyTorch heads (MIOpen JIT makes the first call slow)")
    ap.add_argument("--attn-variant", type=int, default=0)
    ap.add_argument("--aux", action="store_true", help="depth + camera tokens on every view (BASELINE configs[2] with --views 16)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N > 1: torch.distributed backend (nccl = RCCL; gloo only for debugging, e.g. several ranks on one GPU)")
    ap.add_argument("--shard-mode", default="auto", choices=["auto", "heads", "allgather"],
                    help="N > 1: exchange form of the global attention (sharding.ViewSharding)")
    ap.add_argument("--partial-aux", action="store_true", help="cameras on the even views, depth on the second half of the views "
                    "(BASELINE configs[4] with --views 128 --dtype f16)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (one process per GPU)" % args.gpus)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    L.require_gpu()
    if os.environ.get("OVG_FORCE_DEVICE"):      # debugging aid: several ranks on one GPU (only if the RCCL build accepts it)
        local = int(os.environ["OVG_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:                                     # host backend: ViewSharding stages its collectives through the host
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    dtype = DT[args.dtype]
    with torch.device("meta"):
        model = OmniVGGT(compute_dtype=dtype)
    manifest = weights.manifest_of(model)
    sd = weights.synthetic_state_dict(manifest, seed=2)
    model = model.to_empty(device="cpu")
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    agg = model.aggregator
    agg.attn_variant = args.attn_variant
    if world > 1:
        from omnivggt_official_amd.sharding import ViewSharding
        agg.shard = ViewSharding(gather_output=False, mode=args.shard_mode)
    shard_note = {}

    def measure(S, steps, warmup):
        """Time `steps` aggregator forwards on S views; returns the result dict (rank-reduced)."""
        inp = synthetic_inputs(S, dev, aux=args.aux or args.partial_aux)
        idx = list(range(S)) if args.aux else []
        didx, cidx = (list(range(S // 2, S)), list(range(0, S, 2))) if args.partial_aux else (idx, idx)
        n_local = S // world + (1 if rank < S % world else 0)
        nq_local, nk_total = n_local * P_TOK, S * P_TOK
        agg.enable_attention_events(steps * agg.depth)   # live HIP-event timing of the global-attention launches

        def step():
            return agg(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], didx, cidx)

        def barrier():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        # N > 1, automatic exchange form: before timing anything, run the K/V all-gather form and the head-parallel
        # all-to-all form once each on this workload and compare (ViewSharding.choose_mode); all ranks agree on which
        # one may be used, and the JSON line says which one was timed (never part of `value`)
        if dist is not None and args.shard_mode == "auto" and args.dtype != "f32":
            shard_note.update(agg.shard.choose_mode(lambda: step()[0][-1], S))
        for _ in range(warmup):
            step()
        agg.reset_attention_events()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        attn_ms = agg.attention_event_times()
        agg.disable_attention_events()
        attn_avg_ms = sum(attn_ms) / max(len(attn_ms), 1)
        if dist is not None:
            t = torch.tensor([dt, attn_avg_ms], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, attn_avg_ms = float(t[0].item()), float(t[1].item())
        f_total, f_ga = agg_flops(S)
        launch_flops = 4.0 * nq_local * nk_total * 1024           # one global-attention launch on this rank
        achieved = launch_flops / (attn_avg_ms * 1e-3) / 1e12 if attn_avg_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.dtype]
        cfg = "2" if (args.aux and S == 16) else ("4" if (args.partial_aux and S == 128) else ("-" if (args.aux or args.partial_aux) else {8: "1", 64: "3"}.get(S, "-")))
        res = {
            "value": round(S * steps / dt, 3), "ms_per_step": round(dt / steps * 1e3, 3),
            "config": {"workload": "OmniVGGT aggregator forward, %d views 518x518 %s (BASELINE configs[%s]), view-sharded over %d GPU(s)"
                                   % (S, "+ depth + camera tokens" if args.aux else ("+ partial aux (cameras on even views, depth on the second half)" if args.partial_aux else "images-only"), cfg, world),
                       "views": S, "views_per_gpu": n_local, "tokens": S * P_TOK, "weights": "seeded synthetic (no checkpoint offline)",
                       "parallelism": "view-shard x%d%s" % (world, "" if world == 1 else
                                                            (", " + ("head-parallel all-to-all" if (args.shard_mode != "allgather" and S % world == 0 and 16 % world == 0
                                                                                                    and args.dtype != "f32") else "K/V all-gather")))},
            "algorithmic_tflop_per_step": round(f_total / 1e12, 2),
            "tflops_per_gpu": round(f_total / 1e12 / (dt / steps) / world, 1),
            "roofline": {"bound": "mfma", "kernel": ({"bf16": "attn16_kernel<bf16,QB=4,WAVES=4,MODE=0> (speculative anchored softmax + verified fallback)",
                                                       "f16": "attn16_kernel<f16,QB=4,WAVES=4,MODE=1> (lazy-rescale online softmax)"}.get(args.dtype, "attn_kernel<float,1>"))
                         + " (global cross-view attention, D=64)", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": None, "flop_per_launch": launch_flops,
                         "avg_launch_ms": round(attn_avg_ms, 4), "launches_timed": len(attn_ms)},
        }
        del inp
        torch.cuda.empty_cache()
        return res

    S = args.views or 64
    primary = measure(S, args.steps, args.warmup)
    result = {"metric": "frames/sec (518^2, S views) aggregator hot path", "value": primary["value"], "unit": "frames/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": primary["ms_per_step"],
              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic"}
    result.update({k: primary[k] for k in ("config", "algorithmic_tflop_per_step", "tflops_per_gpu", "roofline")})
    if shard_note:
        result["config"]["parallelism"] = "view-shard x%d, %s" % (world, shard_note.get("exchange", "?"))
        result["shard_selfcheck"] = shard_note
    if world == 1 and S != 8 and not args.views:
        sec = measure(8, 10, 3)                                      # BASELINE configs[1] on the same process
        result["secondary"] = {"frames_per_s": sec["value"], "ms_per_step": sec["ms_per_step"], "config": sec["config"],
                               "tflops_per_gpu": sec["tflops_per_gpu"], "roofline": sec["roofline"]}

    if rank == 0 and world == 1:
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr):
            try:
                traffic = json.load(open(tr))
                result["roofline"]["traffic"] = traffic.get("global_attn_S%d_bytes_per_launch" % S)
                if "secondary" in result:
                    result["secondary"]["roofline"]["traffic"] = traffic.get("global_attn_S8_bytes_per_launch")
            except Exception:
                pass
        if args.e2e:
            # whole OmniVGGT.forward (aggregator + camera head + the two DPT heads) on the 8-view config; in the
            # 16-bit modes the DPT heads run on the HIP kernels (heads_hip.py), `--torch-heads` forces PyTorch's
            try:
                Se = args.e2e_views
                inp = synthetic_inputs(Se, dev, aux=args.aux)
                idx = list(range(Se)) if args.aux else []
                model.hip_heads = not args.torch_heads
                full = lambda: model(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], idx, idx)
                full()
                full()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    full()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / 3 * 1e3
                result["e2e"] = {"views": Se, "frames_per_s": round(Se / ms * 1e3, 3), "ms_per_forward": round(ms, 3),
                                 "dpt_heads": "pytorch-f32" if (args.torch_heads or args.dtype == "f32") else "hip-" + args.dtype,
                                 "camera_head": "pytorch-f32"}
            except Exception as e:  # never let the heads hide the hot-path number
                result["e2e_error"] = repr(e)[:200]
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sd)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(sd, S=2):
    """Oracle (CPU restatement, bit-exact vs the reference's PyTorch CPU path) on the host cores:
    a bounded sample -- the full 24-layer aggregator on S=2 views (~10-30 s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import aggregator_oracle as orc
    # intra-op threads: all cores up to 64 (on the 256-core GPU host, 256 torch threads are slower)
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    inp = orc.synthetic_inputs(S)
    with torch.no_grad():
        t0 = time.perf_counter()
        orc.aggregator_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], [], [])
        dt = time.perf_counter() - t0
    return {"value": round(S / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "oracle aggregator forward (fp32, torch CPU, %d threads), S=%d views 518^2 images-only, 1 run, %.1f s" % (cores, S, dt)}


if __name__ == "__main__":
    main()
