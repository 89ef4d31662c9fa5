#!/usr/bin/env python
"""bench.py - guidance maps/sec of the DiG diffusion-guidance forward on MI355X.

One "step" = one pass of the hot path over one batch of synthetic images already resident in HBM:
q_sample at t=100 -> Unet eval forward -> five guidance maps concatenated to [B,1024,H/16,W/16]
(BASELINE.json configs[1]: batch 32, 256x256, bf16 storage / fp32 accumulate).  With N > 1 ranks (one process per
GPU, launched by torch.distributed.run) every rank runs its own batch of 32 (weak scaling) and the step ends with
the single RCCL all-gather of the guidance maps that the north-star names (issued asynchronously and waited for one
step later, so it overlaps the next step's kernels; the last one is drained before the timer stops).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the dominant kernel (conv3x3 implicit GEMM) - algorithmic FLOPs / HIP-event time, vs 2.5 PFLOP/s
  cpu_baseline : the CPU oracle (oracle/dig_oracle.py, a port of the reference's PyTorch graph) timed on this box's
                 host cores on a bounded sample (B=1 images at the same resolution), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_MAP = {256: 234.94, 224: 179.58, 128: 58.43}
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--t", type=int, default=100)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch images per rank (default, the headline); strong: --batch is the GLOBAL batch, split over the ranks "
                         "(BASELINE.json configs[3]: global batch 32 -> 32/N images per GPU)")
    ap.add_argument("--timesteps", type=int, default=1,
                    help="k > 1: every image runs at t = 0..k-1 as ONE batched forward of k x batch images (the guidance loop's "
                         "per-step shape, model.py:296-314 / configs[3]); a map is still one (image, timestep) forward")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="N = 1 only: initialise RCCL with one rank and run the N > 1 code path (async all-gather, barrier, MAX reduce) - a rehearsal, not a measurement")
    ap.add_argument("--profile-kernels", action="store_true", help="per-kernel-class HIP-event breakdown (stderr)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: everything the libraries print on fd 1 meanwhile (RCCL's version
    # banner, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from dig_amd import ops, synth
    from dig_amd.diffusion import GaussianDiffusion, Unet

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.rehearse_collective
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29555")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)
    torch.set_grad_enabled(False)

    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    S = args.size
    if args.scaling == "strong":
        assert args.batch % world == 0, f"--scaling strong needs the global batch {args.batch} to divide over {world} ranks"
        B_img = args.batch // world
    else:
        B_img = args.batch
    B = B_img * args.timesteps  # rows of the batched forward on this rank
    unet = Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
    synth.fill_state_dict(unet, seed=0)
    gd = GaussianDiffusion(unet, image_size=S, timesteps=1000, sampling_timesteps=150).eval()
    cpu_sd = {k: v.clone() for k, v in gd.state_dict().items()} if rank == 0 else None
    gd = gd.to(dev)
    unet.set_compute_dtype(dt)
    # synthetic inputs, resident in HBM before the timed region; each rank gets its own images
    x0 = synth.synth_images(B_img, S, S, seed=1234 + rank * B_img).to(dev)
    if args.timesteps > 1:  # the same images at t = 0..k-1, each with its own noise, as one batch (dig_amd.guidance does the same)
        x0 = x0.repeat(args.timesteps, 1, 1, 1)
        t = torch.arange(args.timesteps, device=dev).repeat_interleave(B_img)
    else:
        t = torch.full((B,), args.t, dtype=torch.long, device=dev)
    nz = synth.synth_noise(B, S, S, seed=1235 + rank * B).to(dev)
    h = S // 16
    from dig_amd.dist import PipelinedGather
    gather = PipelinedGather((B, 1024, h, h), torch.float32, dev, force=args.rehearse_collective) if multi else None

    # ---- HIP-event instrumentation of the dominant kernel (conv3x3) on the launch stream
    conv_log = []  # (class, flops, start_event, end_event)
    orig_conv = ops.conv3x3
    state = {"on": False}

    def conv_timed(x0_, cw, x1=None, aff=None, up2=False, want_stats=True, groups=8):
        if not state["on"]:
            return orig_conv(x0_, cw, x1=x1, aff=aff, up2=up2, want_stats=want_stats, groups=groups)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig_conv(x0_, cw, x1=x1, aff=aff, up2=up2, want_stats=want_stats, groups=groups)
        e1.record()
        Bc, Hc, Wc, Co = out[0].shape
        if Co == 64 and cw.C0 == 64 and cw.C1 == 0 and Hc % 16 == 0 and Wc % 16 == 0 and out[0].dtype == torch.bfloat16:
            cls = "conv3x3_ws64"  # weight-stationary persistent kernel (conv3x3_ws.hip)
        else:
            cls = "conv3x3_bn128" if Co % 128 == 0 else ("conv3x3_bn64" if Co > 32 else "conv3x3_bn32")
        conv_log.append((cls, 2.0 * Bc * Hc * Wc * Co * (cw.C0 + cw.C1) * 9, e0, e1,
                         2.0 * Bc * Hc * Wc * ((cw.C0 + cw.C1) / (4 if up2 else 1) + Co)))  # algorithmic HBM bytes (bf16/fp32 scaled below)
        return out

    ops.conv3x3 = conv_timed

    def step():
        x_t, v, feats = gd.p_losses(x0, t, noise=nz)
        maps = unet._last_maps  # [B,1024,h,h] fp32, == torch.cat(feats, 1) (model.py:313)
        if multi:
            gather.submit(maps)  # this step's all-gather runs on RCCL's stream under the next step's kernels
        return v, maps

    def sync_all():
        if multi:
            gather.drain()  # the last step's gather is inside the timed region
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        step()
    sync_all()
    # timed region: exactly K steps, nothing but the hot path on the stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    # second pass of K steps with a HIP event pair around every conv3x3 launch (the roofline numbers).  The 86 event
    # records per step cost ~0.9 ms of stream bubbles, so this pass is kept out of `value` and reported beside it.
    state["on"] = True
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed_instr = time.perf_counter() - t1
    state["on"] = False
    # third pass (N = 1 only, a few steps): a HIP event pair around EVERY C-ABI launch, so that the per-class table covers the
    # whole step and can be summed against ms_per_step (what is left over is launch gaps + the torch allocator / copies)
    all_log, step_classes = [], None
    if world == 1 and not multi:
        from dig_amd import _lib as L_
        pend = []

        def trace(fn, kw, phase):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            if phase == 0:
                pend.append(ev)
                return
            name = fn
            if fn == "conv3x3":
                if kw["Cout"] == 64 and kw["C0"] == 64 and kw["C1"] == 0:
                    name = "conv3x3 64->64 (ws64)"
                elif kw["Cout"] % 128 == 0:
                    name = "conv3x3 Cout>=128 (db)"
                else:
                    name = "conv3x3 128->64 (rs128)" if kw["C0"] + kw["C1"] == 128 else "conv3x3 other"
            elif fn in ("la_kvctx", "la_merge", "la_out"):
                name = "linear attention (kvctx+merge+out)"
            elif fn in ("gemm_nt", "pw_res"):
                name = fn + (" @256^2" if kw.get("M", kw.get("rows", 0)) >= B * S * S else "")
            all_log.append((name, pend.pop(), ev))

        n3 = min(args.steps, 5)
        L_.TRACE = trace
        for _ in range(n3):
            step()
        torch.cuda.synchronize()
        L_.TRACE = None
        agg = {}
        for name, e0, e1 in all_log:
            d = agg.setdefault(name, [0.0, 0])
            d[0] += e0.elapsed_time(e1)
            d[1] += 1
        step_classes = {k: {"ms_per_step": round(v[0] / n3, 3), "launches_per_step": round(v[1] / n3, 1)}
                        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    maps_per_s = world * B / (elapsed / args.steps)

    # ---- roofline of the dominant kernel class
    per_cls = {}
    esz = 2 if dt == torch.bfloat16 else 4
    for cls, fl, e0, e1, by in conv_log:
        d = per_cls.setdefault(cls, [0.0, 0.0, 0, 0.0])
        d[0] += fl
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += 1
        d[3] += by * esz / 2
    dom = max(per_cls, key=lambda k: per_cls[k][1]) if per_cls else None
    peak = PEAK_BF16_TFLOPS if dt == torch.bfloat16 else PEAK_F32_TFLOPS
    roof = None
    if dom:
        fl, sec, n, by = per_cls[dom]
        ach = fl / sec / 1e12
        # HBM bytes per launch of this kernel class from the committed PMC passes (tools/pmc_traffic.py; bf16, B=32, 256x256 only)
        traffic = None
        tpath = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic.json") for r in (3, 2, 1)) if os.path.exists(q)), None)  # the newest round's passes
        if tpath and dt == torch.bfloat16 and B == 32 and S == 256:
            with open(tpath) as f:
                traffic = json.load(f)["classes"].get(dom, {}).get("hbm_bytes_per_launch")
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes; " + (os.path.basename(tpath) if tpath else "none") + ")",
                "algorithmic_bytes_per_launch_avg": round(by / n), "launches": n, "avg_launch_ms": round(1e3 * sec / n, 4),
                "measured": f"HIP events on the launch stream around every conv3x3 launch, second pass of {args.steps} steps "
                            f"({1e3 * elapsed_instr / args.steps:.3f} ms/step with the event records in the stream)",
                "flops_per_launch_avg": fl / n,
                "classes": {k: {"tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(1e3 * v[1] / args.steps, 3),
                                "launches_per_step": v[2] // args.steps, "hbm_algorithmic_GBps": round(v[3] / v[1] / 1e9, 1)} for k, v in per_cls.items()}}
    if roof and step_classes:
        ssum = sum(v["ms_per_step"] for v in step_classes.values())
        roof["step_classes"] = {"what": "HIP-event time of every C-ABI launch of one step, by kernel class (third pass, events around each launch)",
                                "classes": step_classes, "sum_ms_per_step": round(ssum, 3),
                                "unattributed_ms_per_step": round(ms_per_step - ssum, 3)}
    gflop = GFLOP_PER_MAP.get(S)
    whole = {"tflops": round(maps_per_s / world * gflop / 1e3, 2), "frac_of_mfma_peak": round(maps_per_s / world * gflop / 1e3 / peak, 4)} if gflop else None

    # ---- CPU baseline: the oracle (a port of the reference graph) on the host cores, bounded sample
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import dig_oracle as O
        # Threads: the GPU box hands a 1-GPU job a 16-CPU share of the host (OMP_NUM_THREADS = 16 there; os.cpu_count() shows
        # the whole machine); the baseline uses that share, and the line states what the host shows beside it.
        host_cpus = os.cpu_count() or 1
        try:
            affinity = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            affinity = host_cpus
        share = int(os.environ.get("OMP_NUM_THREADS", "16") or 16)
        cores = max(1, min(share, affinity, host_cpus))
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        torch.set_num_threads(cores)
        cx, cn, ct = x0[:1].cpu(), nz[:1].cpu(), t[:1].cpu()
        ox, ov, of = O.p_losses_eval(cpu_sd, cx, ct, cn)  # warm-up; doubles as the parity check of image 0 of the timed batch
        gx, gv, gf = gd.p_losses(x0, t, noise=nz)
        torch.cuda.synchronize()
        parity = {"what": "max-abs |HIP - CPU oracle (fp32)| on image 0 of the timed batch, per output (max|oracle| beside it)",
                  "x_t": [float((gx[:1].cpu() - ox).abs().max()), float(ox.abs().max())],
                  "v": [float((gv[:1].cpu() - ov).abs().max()), float(ov.abs().max())]}
        for j, (a_, b_) in enumerate(zip(gf, of)):
            parity[f"map{j}"] = [float((a_[:1].cpu() - b_).abs().max()), float(b_.abs().max())]
        tol = 1e-3 if dt == torch.float32 else 2e-2
        parity["tolerance"] = f"{tol:g} * max(1, max|oracle|)" if dt == torch.bfloat16 else "1e-3"
        parity["ok"] = all(v_[0] <= tol * max(1.0, v_[1]) for k_, v_ in parity.items() if isinstance(v_, list))
        n_done, c0 = 0, time.perf_counter()
        while n_done < 10 and time.perf_counter() - c0 < 15.0:
            O.p_losses_eval(cpu_sd, cx, ct, cn)
            n_done += 1
        csec = (time.perf_counter() - c0) / n_done
        cpu = {"value": round(1.0 / csec, 3), "unit": "maps/s", "cores": cores, "kind": "port",
               "host": {"cpu_model": cpu_model, "cpus_visible": host_cpus, "cpus_in_affinity_mask": affinity,
                        "threads_used": cores, "why": "the job's CPU share on the GPU box (OMP_NUM_THREADS), not the whole machine"},
               "sample": f"{n_done} x (B=1, {S}x{S}, t={args.t}) fp32 forwards of oracle/dig_oracle.py (torch CPU ops), {csec*1e3:.0f} ms each"}

    if rank == 0:
        out = {"metric": "guidance maps/sec (Unet fwd 256x256, bs32)", "value": round(maps_per_s, 2), "unit": "maps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic" + (" (REHEARSAL of the N > 1 code path with one RCCL rank)" if args.rehearse_collective else ""),
               "config": {"workload": (f"BASELINE.json configs[1]: batch-{B} Unet(dim=64, mults 1,2,4,8) guidance extraction at {S}x{S}, "
                                       f"t={args.t}, q_sample + Unet eval forward + 5 guidance maps -> [B,1024,{h},{h}]") if args.timesteps == 1 else
                                      (f"BASELINE.json configs[3]: {B_img} images/GPU x t=0..{args.timesteps - 1} as one batch of {B} Unet(dim=64) "
                                       f"guidance forwards at {S}x{S} -> [{B_img},{args.timesteps},1024,{h},{h}]"),
                          "per_gpu_batch": B, "global_batch": world * B, "images_per_gpu": B_img, "timesteps_batched": args.timesteps,
                          "image": S, "parallelism": f"dp{world}",
                          "collective": "all_gather_into_tensor(maps fp32 [B,1024,h,w] - fp32 because the maps' consumer (LFCA, model.py:322) and the reference's own feat tensors are fp32), overlapped with the next step" if world > 1 else "none"},
               "whole_forward": whole, "roofline": roof, "cpu_baseline": cpu, "parity": parity}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)


if __name__ == "__main__":
    main()
