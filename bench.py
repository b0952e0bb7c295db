#!/usr/bin/env python3
"""Headline benchmark: training tokens/s of the fused ALiBi transformer step + decode tokens/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train_unified|train_separated|
                     train_relative|decode_unified] [--batch B] [--seq S] [--dtype bf16|fp32]

One "step" = one full training step (forward, fused softmax-CE, backward, gradient all-reduce when N>1,
fused AdamW) on one batch of synthetic MIDI-grammar tokens already resident in HBM.  N=1 default workload =
BASELINE.json configs[1]: alibi_unified d=1024 L=6 H=8 ff=4096 V=618, bf16, seq 2048, batch 16 per GPU.
For N>1 launch under torch.distributed.run (one rank per GPU, RCCL); weak scaling.

Rank 0 prints ONE JSON line: the training metric; `roofline` = the kernel with the largest share of the step
(HIP-event timed inside the timed region; every other MFMA kernel in `roofline_other_kernels`); `decode` = the
second half of BASELINE.json's metric (configs[3]: KV-cached greedy sampling, batch 64, 4096-token continuation:
tokens/s, fraction of the HBM roofline for the whole step and for the decode-attention kernel); `cpu_baseline` = the
oracle's restatement of the reference step on the host cores (rank 0, N=1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0         # HBM3E spec, same table
TRAFFIC_FILE = "r03a_traffic_pmc.json"


def pmc_traffic(key):
    """HBM bytes per launch of a roofline kernel from the committed TCC-counter passes (tools/traffic.sh:
    FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, read side doubled as MI355X_MICROARCH.md's HBM
    section prescribes for gfx950).  None when no measurement exists for this exact kernel configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)) as f:
            t = json.load(f)
        return float(t[key]["hbm_bytes_corrected"])
    except (OSError, KeyError, ValueError):
        return None


def fwd_flops_per_token(L, d, ff, S, V, s_eff_factor=1.0):
    """L*(2d*3d + 2d*d + 4d*ff + 2d*S_eff) + 2dV, causal attention at half the square."""
    return L * (2 * d * 3 * d + 2 * d * d + 4 * d * ff + 2 * d * S * s_eff_factor) + 2 * d * V


def build_workload(name, batch, seq, dtype, device):
    from musicgeneration_amd.synthetic import SyntheticSeparatedDataset, SyntheticUnifiedDataset
    rank = int(os.environ.get("RANK", "0"))
    if name == "train_unified":
        from musicgeneration_amd.model_baseline import build_everything_unified as cfg
        args = cfg.m_args.get_copy()
        args.max_sequence = max(seq, args.max_sequence)
        args.dtype = dtype
        model, crit = cfg.build_model_and_criterion(args)
        data = SyntheticUnifiedDataset(batch, seq, seed=1234 + rank, device=device)
        fl = fwd_flops_per_token(args.n_layers, args.embed_dim, args.proj_forward, seq, args.vocab)
        desc = f"alibi_unified d={args.embed_dim} L={args.n_layers} H={args.n_heads} ff={args.proj_forward} V={args.vocab}"
        lr, wd = cfg.opt_args.lr, cfg.opt_args.weight_decay
        shape = (args.n_heads, args.embed_dim // args.n_heads, args.n_layers)
    elif name == "train_separated":
        from musicgeneration_amd.model_baseline import build_everything_separated as cfg
        args = cfg.m_args.get_copy()
        args.max_sequence = max(seq, args.max_sequence)
        args.dtype = dtype
        model, crit = cfg.build_model_and_criterion(args)
        data = SyntheticSeparatedDataset(batch, seq, seed=1234 + rank, device=device)
        d = args.embed_note + args.embed_velocity + args.embed_duration
        fl = fwd_flops_per_token(args.n_layers, d, args.proj_forward, seq, 425)
        desc = f"alibi_separated d={d} L={args.n_layers} H={args.n_heads} ff={args.proj_forward} V=92/130/203"
        lr, wd = cfg.opt_args.lr, cfg.opt_args.weight_decay
        shape = (args.n_heads, d // args.n_heads, args.n_layers)
    elif name == "train_relative":
        from musicgeneration_amd.old_models.relative_transformer import RelativeTransformer
        from musicgeneration_amd.model_baseline.alibi_unified import Criterion
        model = RelativeTransformer(618, 512, 6, 8, max(seq, 4096), 2048, dropout=0.1, dtype=dtype)
        crit = Criterion(None)                                   # plain mean CE (old_models/train.py's loss)
        data = SyntheticUnifiedDataset(batch, seq, seed=1234 + rank, device=device)
        fl = fwd_flops_per_token(6, 512, 2048, seq, 618, 1.5)
        desc = "relative_transformer d=512 L=6 H=8 ff=2048 V=618"
        lr, wd = 1e-3, 1e-4
        shape = (8, 64, 6)
    else:
        raise SystemExit(f"unknown workload {name}")
    return model.to(device), crit.to(device), data, fl, desc, lr, wd, shape


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            info = f.read()
    except OSError:
        return "unknown", None
    model = next((l.split(":", 1)[1].strip() for l in info.splitlines() if l.startswith("model name")), "unknown")
    cores = set()
    phys = None
    for l in info.splitlines():
        if l.startswith("physical id"):
            phys = l.split(":")[1].strip()
        elif l.startswith("core id"):
            cores.add((phys, l.split(":")[1].strip()))
    return model, (len(cores) or None)


def cpu_baseline(steps=5, warmup=2):
    """the oracle (CPU restatement of the reference, parity-pinned by golden vectors) runs config 1
    — unified d=512 L=6 H=8 ff=4096, fp32, batch 8 x seq 1024, dropout 0.1, fwd + loss + bwd + AdamW(1e-3, wd 1e-4):
    2 warm-up + 5 timed steps, median.  Threads: torch's default on the box (its CPU share: OMP_NUM_THREADS), which
    is what `cores` reports; the machine's physical core count and CPU model ride along in `sample`."""
    from oracle import reference_port as O
    threads = torch.get_num_threads()
    model, phys = _cpu_model()
    g = torch.Generator().manual_seed(1234)
    V, d, L, H, ff, B, S = 618, 512, 6, 8, 4096, 8, 1024
    sd = {"embed.weight": torch.randn(V, d, generator=g), "norm.g": torch.ones(d), "norm.b": torch.zeros(d),
          "proj_class.weight": torch.randn(V, d, generator=g) * d ** -0.5, "proj_class.bias": torch.zeros(V)}
    for i in range(L):
        p = f"decoder_layers.{i}."
        sd.update({p + "attn.c_attn.w": 0.02 * torch.randn(1, d, 3 * d, generator=g), p + "attn.c_attn.b": torch.zeros(3 * d),
                   p + "attn.c_proj.w": 0.02 * torch.randn(1, d, d, generator=g), p + "attn.c_proj.b": torch.zeros(d),
                   p + "linear1.weight": torch.randn(ff, d, generator=g) * d ** -0.5, p + "linear1.bias": torch.zeros(ff),
                   p + "linear2.weight": torch.randn(d, ff, generator=g) * ff ** -0.5, p + "linear2.bias": torch.zeros(d),
                   p + "norm1.weight": torch.ones(d), p + "norm1.bias": torch.zeros(d),
                   p + "norm2.weight": torch.ones(d), p + "norm2.bias": torch.zeros(d)})
    sd = O.make_trainable(sd)
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-3, weight_decay=1e-4)
    toks = O.synthetic_unified_tokens(B, S + 1, seed=1234)
    x, y = toks[:, :-1], toks[:, 1:]
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        O.unified_train_step(sd, opt, x, y, H, pad_index=1, p_drop=0.1)
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B * S / med, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"oracle fp32 train step, unified d=512 L=6 S=1024 batch 8 (configs[0]), dropout 0.1, AdamW; "
                      f"{steps} timed steps after {warmup} warm-up, median {med:.2f} s/step; {threads} torch threads on "
                      f"'{model}' (os.cpu_count()={os.cpu_count()}, physical cores={phys})"}


DEFAULT_SHAPES = {"train_unified": (16, 2048), "train_separated": (8, 1024), "train_relative": (8, 4096)}


def bench_training(a, workload, B, S, steps, warmup, device, rank, world, force_dp):
    """One training workload: `warmup` untimed steps, then exactly `steps` timed ones between barriers; returns the
    record (tokens/s over all ranks, step time, roofline lines of the MFMA kernels from HIP-event pairs)."""
    from musicgeneration_amd import ops
    from musicgeneration_amd.optim import FusedAdamW
    torch.manual_seed(4321)                              # reproducible init and dropout seeds (rank 0's weights are broadcast)
    model, crit, data, fwd_fl, desc, lr, wd, (H, dh, L) = build_workload(workload, B, S, a.dtype, device)
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=wd)
    ddp = None
    if world > 1 or force_dp:
        from musicgeneration_amd.data_parallel import GradientBucketer
        ddp = GradientBucketer(model, mode=a.dp_mode)
    batches = [data.get() for _ in range(4)]          # resident in HBM before the timed region
    model.train()

    def step(i):
        x, y = batches[i % len(batches)]
        loss, _ = crit(model(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if ddp is not None:
            ddp.finish()
        opt.step()
        return loss

    for i in range(warmup):
        step(i)
    # The objects built so far (model, golden-size module trees, autograd graphs of the warm-up) leave the garbage
    # collector's generations: a full collection over them takes ~90 ms on the host (tools/dp_probe.py: one such pause
    # in the 7th step of a data-parallel run, whose bucket bookkeeping allocates the most containers per step), and
    # early in a run the host is not yet far enough ahead of the GPU to hide it.
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP-event pairs around the MFMA kernels' launches, inside the timed region, on the first EVENT_STEPS steps
    # only (~100 event pairs per step cost ~0.2 ms of the step they bracket)
    EVENT_STEPS = min(3, steps)
    events = ops.KERNEL_EVENTS = {}
    t0 = time.perf_counter()
    for i in range(steps):
        if i == EVENT_STEPS:
            ops.KERNEL_EVENTS = None
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.KERNEL_EVENTS = None
    gc.unfreeze()
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    final_loss = float(loss.detach())

    tokens = world * B * S * steps
    value = tokens / dt
    train_fl = 3 * fwd_fl
    # roofline: every MFMA kernel of the step is one of libmgx's and is timed by a HIP-event pair around its launches
    # inside the timed region (musicgeneration_amd.ops._timed, on the stream the kernels run on); the headline
    # `roofline` object is the one with the largest total time.  achieved = algorithmic FLOPs of the launches / their time.
    labels = {"alibi_attn_fwd": "attn_fwd_mfma_kernel (ALiBi flash forward; QK^T + PV at half the square)",
              "rel_attn_fwd": "rel_fwd_mfma_kernel (skew relative flash forward; QK^T + QE^T + PV at half the square)",
              "rel_attn_bwd": "delta + rel_bwd_dkv + rel_bwd_dq (stores dG) + dE GEMM over the dG blocks + reduce (6 products "
                              "at half the square; 12 executed: S, dP and Q.E^T are recomputed in the dK/dV and dQ kernels)",
              "alibi_attn_bwd": "attn_bwd_stats_kernel + attn_bwd_dkv_pipe_kernel + attn_bwd_dq_pipe_kernel (5 products at half the square)",
              "gemm_nt": "gemm_nt_kernel (every x.W and dy.W^T projection of the blocks and the head: forward + dgrad, "
                         "bias / residual / activation-backward epilogues)",
              "gemm_tn": "gemm_tn_kernel + splitk_reduce_kernel (weight gradients of the blocks, two per launch, split-K fp32 slabs)",
              "gemm_tn_head": "gemm_tn_kernel + splitk_reduce_kernel (head weight gradient)",
              "gemm_tn_embed": "gemm_tn_kernel + splitk_reduce_kernel (embedding gradient as a one-hot GEMM)"}
    lines = []
    for name, ev in events.items():
        if name not in labels or not ev:
            continue
        tot_ms = sum(s.elapsed_time(e) for s, e, _ in ev)
        fl = sum(w for _, _, w in ev)
        ach = fl / (tot_ms * 1e-3) / 1e12
        traffic = None
        if workload == "train_unified" and a.dtype == "bf16" and (B, S, H, dh) == (16, 2048, 8, 128):   # d=1024 ff=4096 fixed by the workload
            traffic = pmc_traffic(name)
        lines.append({"bound": "mfma", "kernel": labels[name], "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                      "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                      "traffic_unit": f"HBM bytes per launch (TCC FETCH_SIZE x2 + WRITE_SIZE, profiles/{TRAFFIC_FILE})",
                      "launches_timed": len(ev), "avg_launch_ms": tot_ms / len(ev), "flops_per_launch": fl / len(ev),
                      "ms_per_step": tot_ms / EVENT_STEPS})
    lines.sort(key=lambda r: -r["ms_per_step"])
    roof = lines[0] if lines else None
    # the HBM-bound kernels of the step, from the same event pairs: algorithmic bytes / time against the 8 TB/s peak
    hbm_labels = {"ln_fwd": "ln_fwd_kernel ((dropout +) residual + LayerNorm forward)",
                  "ln_bwd": "ln_bwd_kernel (LayerNorm backward + bias-gradient partials) + column_reduce_kernel",
                  "adamw": "adamw_multi_kernel (28 B per parameter)"}
    hbm_lines = []
    for name, ev in events.items():
        if name not in hbm_labels or not ev:
            continue
        tot_ms = sum(s.elapsed_time(e) for s, e, _ in ev)
        by = sum(w for _, _, w in ev)
        ach = by / (tot_ms * 1e-3) / 1e9
        hbm_lines.append({"bound": "hbm", "kernel": hbm_labels[name], "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": ach / PEAK_HBM_GBS, "launches_timed": len(ev), "avg_launch_ms": tot_ms / len(ev),
                          "bytes_per_launch": by / len(ev), "ms_per_step": tot_ms / EVENT_STEPS})
    if ddp is not None:
        ddp.close()                                      # the next workload builds its own bucketer (and CE hook)
    out = {
        "metric": "training tokens/sec", "value": value, "unit": "tokens/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic MIDI-grammar tokens, random-init weights",
        "config": {"workload": f"{workload}: {desc}, seq {S}, batch {B}/GPU, dropout 0.1, fused AdamW",
                   "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}"},
        "step_tflops": value * train_fl / 1e12, "step_frac_of_bf16_peak": value * train_fl / 1e12 / PEAK_BF16_TFLOPS / world,
        "final_loss": final_loss,
        "roofline": roof,
        "roofline_other_kernels": lines[1:],
        "hbm_bound_kernels": hbm_lines,
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="train_unified")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch")
    ap.add_argument("--seq", type=int, default=None)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode half of the metric")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the train_separated / train_relative sub-records of the default run")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel path (RCCL group, gradient bucketer) with one rank: its test on a one-GPU box")
    ap.add_argument("--dp-mode", default="flat", choices=["flat", "inplace"], help="GradientBucketer exchange mode")
    a = ap.parse_args()

    # stdout carries ONE line, the JSON record: anything libraries print on the way (RCCL writes its version banner to
    # stdout when the communicator is created) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(record):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(record), flush=True)
        os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (musicgeneration_amd has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    force_dp = world == 1 and a.force_dp
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    if a.workload == "decode_unified":
        out = bench_decode(a, device, rank, world, a.batch or 64, a.seq or 4096)
        if rank == 0:
            emit(out)
        if world > 1:
            dist.destroy_process_group()
        return

    B = a.batch or DEFAULT_SHAPES[a.workload][0]
    S = a.seq or DEFAULT_SHAPES[a.workload][1]
    out = bench_training(a, a.workload, B, S, a.steps, a.warmup, device, rank, world, force_dp)
    torch.cuda.empty_cache()
    if a.workload == "train_unified" and not a.no_decode:
        # the other half of BASELINE.json's metric, in the same driver-visible record
        out["decode"] = bench_decode(a, device, rank, world, 64, 4096)
    if a.workload == "train_unified" and not a.no_other_configs and a.batch is None and a.seq is None:
        # BASELINE.json's other training configurations (configs[2] and configs[4]) in the same driver-visible record:
        # fewer steps, same measurement
        out["other_configs"] = {}
        for wl in ("train_separated", "train_relative"):
            torch.cuda.empty_cache()
            out["other_configs"][wl] = bench_training(a, wl, *DEFAULT_SHAPES[wl], min(a.steps, 6), min(a.warmup, 2), device,
                                                      rank, world, force_dp)
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        emit(compact_record(out, a))
    if world > 1 or force_dp:
        dist.destroy_process_group()


def _short(line):
    """One roofline line without its prose: what a reader of the stored record needs to recompute the fraction."""
    if line is None:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_timed", "ms_per_step",
            "flops_per_launch", "bytes_per_launch")
    r = {"kernel": line["kernel"].split(" (")[0]}
    r.update({k: (round(line[k], 6) if isinstance(line[k], float) else line[k]) for k in keep if k in line})
    return r


def compact_record(out, a):
    """The ONE line the driver stores keeps only the tail of a long line: the full record (every kernel's prose label,
    the per-kernel lists of the other configurations) goes to gpurun_out/bench_full.json, and the printed line is
    ordered so that what BASELINE.json's metric names comes LAST: other_configs (value / ms_per_step / step fraction /
    headline roofline each), then `decode`, then `cpu_baseline`."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    rec = {k: v for k, v in out.items() if k not in ("roofline_other_kernels", "hbm_bound_kernels", "other_configs",
                                                       "decode", "cpu_baseline")}
    if "roofline_other_kernels" in out:
        rec["roofline_other_kernels"] = [_short(l) for l in out["roofline_other_kernels"]]
    if "hbm_bound_kernels" in out:
        rec["hbm_bound_kernels"] = [_short(l) for l in out["hbm_bound_kernels"]]
    if "other_configs" in out:
        rec["other_configs"] = {
            wl: {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                 "workload": r["config"]["workload"], "step_frac_of_bf16_peak": r["step_frac_of_bf16_peak"],
                 "final_loss": r["final_loss"], "roofline": _short(r["roofline"]),
                 "roofline_other_kernels": [{"kernel": l["kernel"].split(" (")[0], "frac": round(l["frac"], 4),
                                             "ms_per_step": round(l["ms_per_step"], 4)} for l in r["roofline_other_kernels"]]}
            for wl, r in out["other_configs"].items()}
    rec["full_record"] = "gpurun_out/bench_full.json (per-kernel lists with labels; profiles/ holds the builder's copy)"
    for k in ("decode", "cpu_baseline"):
        if k in out:
            rec[k] = out[k]
    return rec


def bench_decode(a, device, rank, world, B, new):
    """configs[3]: KV-cached greedy sampling, unified d=1024 L=6, batch 64/GPU, 4096-token continuation
    (shortened by --seq for quick runs).  Returns the decode record: generated tokens/s (whole job), the step's
    fraction of the HBM roofline (SURVEY 8(d) bytes: weights once + K/V rows of every layer at the mean context)
    and the HBM roofline of the decode-attention kernel (HIP-event timed in a launch-by-launch pass)."""
    from musicgeneration_amd.model_baseline import build_everything_unified as cfg
    from musicgeneration_amd.decode import sample_next_plus_one
    from musicgeneration_amd import ops
    L = 64 + new + 1
    args = cfg.m_args.get_copy()
    args.max_sequence = L
    args.dtype = a.dtype
    torch.manual_seed(4321)
    model, _ = cfg.build_model_and_criterion(args)
    model = model.to(device).eval()
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    seq = torch.zeros(B, L, dtype=torch.int32, device=device)
    seq[:, 1:64] = torch.randint(2, 618, (B, 63), generator=g, device=device, dtype=torch.int32)
    warm = seq[:, :80].clone()
    sample_next_plus_one(model, warm, idx=64, deterministic=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    seq0 = seq.clone()
    gc.collect()
    gc.freeze()                                                           # see bench_training
    t0 = time.perf_counter()
    sample_next_plus_one(model, seq, idx=64, deterministic=True)          # HIP-graph replay per step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # the stochastic branch (inference_unified.py:16-17; SURVEY 8(d) Cfg 4 "greedy + multinomial"): same graph with the
    # counter-based Gumbel-max draw in the head kernel
    seq_s = seq0.clone()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    sample_next_plus_one(model, seq_s, idx=64, deterministic=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt_s = time.perf_counter() - t1
    gc.unfreeze()
    # second, untimed-for-`value` pass over the head of the same continuation, launch by launch, with a HIP-event
    # pair around every decode-attention launch (a graph replay cannot be bracketed per kernel)
    ev_new = min(new, 1024)
    seq1 = seq0[:, :64 + ev_new + 1].clone()
    ops.KERNEL_EVENTS = {}
    sample_next_plus_one(model, seq1, idx=64, deterministic=True, use_graph=False)
    torch.cuda.synchronize()
    events, ops.KERNEL_EVENTS = ops.KERNEL_EVENTS, None
    same = bool(torch.equal(seq1[:, :-1], seq[:, :64 + ev_new]))
    if world > 1:
        t = torch.tensor([dt, dt_s], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_s = float(t[0]), float(t[1])
    steps = L - 1 - 64
    value = world * B * steps / dt
    esz = 2 if a.dtype == "bf16" else 4
    d, Ln = args.embed_dim, args.n_layers
    ev = events.get("decode_attn")
    roof = None
    if ev:
        ms = sum(s.elapsed_time(e) for s, e, _ in ev) / len(ev)
        mean_ctx = 64 + (ev_new - 1) / 2
        by = B * 2 * d * mean_ctx * esz                     # K and V rows of one layer, mean context of the timed launches
        ach = by / (ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "decode_attn_bf16_kernel (append + attend, 3 query rows share every K/V load)",
                "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                "traffic": pmc_traffic("decode_attn") if (B, a.dtype) == (64, "bf16") else None,
                "traffic_unit": f"HBM bytes per launch at context 2112 (TCC FETCH_SIZE x2 + WRITE_SIZE, profiles/{TRAFFIC_FILE})",
                "launches_timed": len(ev), "avg_launch_ms": ms, "bytes_per_launch": by,
                "timed_in": f"launch-by-launch pass over the first {ev_new} steps of the same continuation (mean context "
                            f"{mean_ctx:.0f})", "graph_tokens_equal": same}
    # whole-step HBM roofline of SURVEY 8(d): weights read once per step + the K/V rows of every layer at the mean context
    w_bytes = esz * (sum(p.numel() for l in model.decoder_layers for p in l.parameters()) + model.proj_class.weight.numel())
    step_bytes = w_bytes + B * 2 * Ln * d * (64 + steps / 2) * esz
    step_gbs = step_bytes / (dt / steps) / 1e9
    return {"metric": "decode tokens/sec", "value": value, "unit": "tokens/s", "n_gpus": world, "steps": steps, "warmup": 16,
            "step_bytes": step_bytes, "step_gb_per_s": step_gbs, "step_frac_of_hbm_peak": step_gbs / PEAK_HBM_GBS,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "stochastic": {"value": world * B * steps / dt_s, "unit": "tokens/s", "ms_per_step": dt_s / steps * 1e3,
                           "draw": "deterministic=False: Gumbel-max over softmax(logits) in the head kernel, counter-based RNG, "
                                   "same HIP graph", "tokens_in_vocab": bool(int(seq_s.max()) < 618 and int(seq_s.min()) >= 0)},
            "dtype": a.dtype, "data": "synthetic primer, random-init weights",
            "config": {"workload": f"decode_unified: alibi_unified d={d} L={Ln}, batch {B}/GPU, {steps}-token greedy "
                                   f"continuation from a 63-token primer, KV cache", "global_batch": world * B,
                       "seq_len": L, "parallelism": f"replicas{world}"},
            "roofline": roof}


if __name__ == "__main__":
    main()
