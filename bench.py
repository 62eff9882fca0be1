"""Headline benchmark: ViT-L video encoder forward+backward, 16 x 256 x 256 clips, batch 32 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic clips already resident in HBM: encoder
forward, loss = mean(out^2), encoder backward to every parameter gradient and, for N > 1, the bucketed
gradient all-reduce (RCCL) overlapped with backward.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel is the 256x256-tile bf16 MFMA GEMM (plain bf16 epilogue); its launches are
bracketed with HIP events on the launch stream during the timed steps; achieved = algorithmic FLOPs
(2*M*N*K per launch) / measured time, against the 2.5 PFLOP/s dense bf16 peak.
cpu_baseline: the CPU oracle (a from-scratch port of the reference arithmetic, fp32) on this host's
cores for ONE clip forward+backward of the same model, same weights, same clip -- rank 0, N = 1 only; the
device's output for that clip is compared with the oracle's (relative L2) in the same leg.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_FLOPS = 2.5e15      # dense bf16 MFMA peak of one MI355X
DEPTH, HEADS, DIM = 24, 16, 1024
FRAMES, IMG, BATCH = 16, 256, 32


def flops_per_clip(n_tokens: int) -> float:
    per_tok = 2 * DIM * 3 * DIM + 2 * DIM * DIM + 4 * DIM * 4 * DIM + 4 * n_tokens * DIM
    fwd = DEPTH * n_tokens * per_tok + 2 * 1536 * DIM * n_tokens
    return 3.0 * fwd


class GemmProbe:
    """HIP-event brackets around every launch of the dominant GEMM kernel while enabled."""

    def __init__(self, ops):
        self.ops, self.enabled, self.records = ops, False, []
        self._orig = ops.gemm_nt

    def install(self):
        def wrapped(x, w, epilogue, out, out2=None, bias=None, resid=None, aux=None, tile=0, **kw):
            M, K = x.shape
            N = w.shape[0]
            dominant = self.enabled and epilogue == self.ops.EPI_BF16 and bias is None and N % 256 == 0 and M >= 2048 \
                and tile in (0, 256, 4256)
            if not dominant:
                return self._orig(x, w, epilogue, out, out2, bias, resid, aux, tile, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self._orig(x, w, epilogue, out, out2, bias, resid, aux, tile, **kw)
            e.record()
            self.records.append((2.0 * M * N * K, s, e))
            return r
        self.ops.gemm_nt = wrapped
        import vit_amd.encoder_engine as eng
        eng.ops.gemm_nt = wrapped

    def summary(self):
        if not self.records:
            return None
        fl = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return {"launches": len(self.records), "flops": fl, "seconds": ms * 1e-3}


def cpu_baseline(enc, clip, y_gpu):
    """One ViT-L clip forward+backward through the CPU oracle (fp32, all host threads torch uses) on the SAME weights
    and the same clip the device model just ran: besides the timing, the relative L2 distance between the device's
    output for that clip and the oracle's is reported."""
    from oracle import ref_cpu as R
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in enc.state_dict().items() if "inv_freq" not in k}
    t0 = time.perf_counter()
    y = R.encoder_forward(P, clip, DEPTH, HEADS)
    (y ** 2).mean().backward()
    dt = time.perf_counter() - t0
    err = ((y_gpu - y.detach()).norm() / y.detach().norm()).item()
    return {"value": 1.0 / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 clip (16x256x256, 2048 tokens) forward+backward through all {DEPTH} blocks, fp32, {dt:.1f} s",
            "device_vs_oracle_output_rel_l2": round(err, 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="clips per GPU (the headline number uses 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-mode", action="store_true",
                    help="model.train(): the blocks' p=0.1 dropouts are applied (the headline number is eval mode)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1 or world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: for N > 1 launch one rank per GPU with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {max(args.gpus, 1)} --master-addr 127.0.0.1 "
                         f"--master-port P bench.py --gpus {max(args.gpus, 1)} ...`")
    if not (0 <= rank < world and 0 <= local_rank < world):
        raise SystemExit(f"bench.py: RANK={rank} / LOCAL_RANK={local_rank} outside a world of {world} (one node, one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from vit_amd import ops
    from vit_amd import surgical_world_model as swm
    from vit_amd.ddp import BucketReducer

    torch.manual_seed(42)
    enc = swm.VideoViTEncoder(img_size=IMG, num_frames=FRAMES, embed_dim=DIM, depth=DEPTH, num_heads=HEADS).eval().to(dev)
    if args.train_mode:
        enc.train()
    reducer = None
    if world > 1:
        for p in enc.parameters():                       # identical replicas
            dist.broadcast(p.data, src=0)
        reducer = BucketReducer()
        enc.set_grad_reducer(reducer)
        if dist.get_world_size() != world or dist.get_backend() != "nccl":
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks on backend {dist.get_backend()!r}, "
                             f"expected {world} on 'nccl' (RCCL)")
    video = torch.randn(args.batch, 3, FRAMES, IMG, IMG, generator=torch.Generator().manual_seed(1234 + rank)).to(dev)
    n_tokens = (FRAMES // 2) * (IMG // 16) * (IMG // 16)

    probe = GemmProbe(ops)
    probe.install()

    def step():
        enc.zero_grad(set_to_none=True)
        enc._engine.weights.invalidate()      # as after an optimizer update: bf16 operand copies are re-made
        y = enc(video)
        loss = (y ** 2).mean()
        loss.backward()
        return loss

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    probe.enabled = True
    if reducer is not None:
        reducer.buckets_reduced = reducer.bytes_reduced = 0
        reducer.measure_exposed = True
        reducer.exposed_ms()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    probe.enabled = False
    comm = None
    if world > 1:
        exposed = reducer.exposed_ms()                   # one join per step, after the timed region (it synchronises)
        t = torch.tensor([elapsed, sum(exposed) / max(len(exposed), 1), max(exposed or [0.0])], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t[0].item()
        # what the data-parallel exchange was and what it cost, so that a scaling run explains itself
        comm = {"backend": "nccl (RCCL)", "ranks_in_process_group": dist.get_world_size(),
                "buckets_per_step": reducer.buckets_reduced // args.steps,
                "bytes_reduced_per_step": reducer.bytes_reduced // args.steps, "reduce_op": "avg, f32, one bucket per block",
                "exposed_comm_ms_per_step_max_over_ranks": round(t[1].item(), 3),
                "exposed_comm_ms_worst_step": round(t[2].item(), 3),
                "exposed_comm_definition": "device time between the end of the last backward kernel and the end of the last "
                                           "bucket's all-reduce (HIP events on the compute stream around the join)"}

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / elapsed
        ps = probe.summary()
        roofline = None
        # HBM bytes per launch of the dominant kernel come from a separate counter run (PMC passes cannot share
        # a run with the timed region); the summary that rocprofv3 run produced is kept next to the profiles.
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and args.batch == BATCH:
            with open(tpath) as f:
                traffic = json.load(f).get("traffic_bytes_per_launch")
        if ps:
            achieved = ps["flops"] / ps["seconds"] / 1e12
            roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_FLOPS / 1e12,
                        "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_BF16_FLOPS, 4), "traffic": traffic,
                        "kernel": "gemm_nt_quad_kernel<bf16-out> (256x256 tile, 4 waves, persistent)", "launches": ps["launches"],
                        "avg_launch_ms": round(ps["seconds"] * 1e3 / ps["launches"], 4)}
        out = {
            "metric": "video clips/sec (ViT-L fwd+bwd, 16x256^2)", "value": round(value, 3), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ViT-L encoder (dim 1024, depth 24, heads 16) forward+backward, 16x256x256 clips, "
                                   "tubelet 2x16x16 (2048 tokens/clip), "
                                   + ("training-mode dropout p=0.1" if args.train_mode else "eval-mode dropout")
                                   + ", loss = mean(out^2)",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "model_flops_utilization": round(value * flops_per_clip(n_tokens) / (world * PEAK_BF16_FLOPS), 4),
            "loss": float(loss.detach()),
            "roofline": roofline,
        }
        if comm is not None:
            out["extra"] = {"data_parallel": comm}
        if world == 1 and not args.no_cpu_baseline:
            enc.eval()                                   # the oracle's arithmetic is the eval-mode one
            with torch.no_grad():
                y_gpu = enc(video[:1].contiguous()).cpu()
            out["cpu_baseline"] = cpu_baseline(enc, video[:1].cpu(), y_gpu)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
