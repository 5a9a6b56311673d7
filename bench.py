"""bench.py - images/s (whole job) for the 768x768, 50-DDIM-step, classifier-free-guided SD-2 inpainting hot path
(UNet step x50 + fused CFG/DDIM update + AutoencoderKL decode) on N MI355X GPUs, one process per GPU.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one batch of images generated end to end (configs[1] of BASELINE.json: 768^2, batch 1 per GPU, 50 steps,
CFG).  Inputs (synthetic) and random-init checkpoint-shaped weights are resident in HBM before the timed
region.  Images are sharded one batch per GPU with no communication during sampling and ONE all-gather of the uint8
outputs per step (weak scaling).  Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the fp32 oracle timed on the host cores, rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import stable_diffusion_amd  # noqa: E402

stable_diffusion_amd.install_dropin()
from stable_diffusion_amd import _lib, engine as E  # noqa: E402
from stable_diffusion_amd.sampler import Sampler  # noqa: E402
from stable_diffusion_amd.shard import all_gather_images  # noqa: E402
from stable_diffusion_amd.synth import fill_state_dict, make_inputs  # noqa: E402

PEAK_MFMA_TFLOPS = 2500.0   # dense bf16/fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
# algorithmic FLOPs (2*MAC over conv + linear + attention), SURVEY.md 8(a)/BASELINE.md section 3, per UNet forward at
# CFG batch 2 and per decode at batch 1
UNET_GFLOP = {64: 1608.8, 96: 4298.7, 128: 9350.9}
DECODE_GFLOP = {64: 2514.5, 96: 5754.3, 128: 10470.4}


def work_of(name, a):
    """(algorithmic flops, algorithmic bytes) of one C-ABI call from its arguments (see include/sdmi.h).  `name` is the
    kernel symbol for conv/gemm/attention calls (sdmi_last_kernel), the entry point otherwise."""
    if name.startswith("conv_win") or (name.startswith("igemm_kernel") and "true," in name):
        n, cin, cout, ho, wo = a[2], a[5], a[7], a[8], a[9]
        return 2.0 * n * ho * wo * cout * 9 * cin, 2.0 * (n * a[3] * a[4] * cin + n * ho * wo * cout + 9 * cin * cout)
    if name.startswith("igemm_kernel"):
        m, n, k, b = a[9], a[10], a[11], a[12]
        return 2.0 * m * n * k * b, 2.0 * b * (m * k + n * k + m * n)
    if name.startswith("attn64"):
        b, h, nq, nk = a[12], a[13], a[14], a[15]
        return 4.0 * b * h * nq * nk * 64, 2.0 * b * h * 64 * (2 * nq + 2 * nk)
    if name == "sdmi_groupnorm":
        return 0.0, 4.0 * a[4] * a[5] * a[6]
    if name == "sdmi_groupnorm_from_colstats":      # fold of the producer's statistics + apply: the tensor read once, written once
        return 0.0, 4.0 * a[4] * a[5] * a[6]
    if name == "sdmi_groupnorm_stats":
        return 0.0, 2.0 * a[1] * a[2] * a[3]
    if name == "sdmi_groupnorm_apply":
        return 0.0, 4.0 * a[5] * a[6] * a[7]
    if name == "sdmi_layernorm":
        return 0.0, 4.0 * a[7] * a[8]
    return 0.0, 0.0


def build_models(dev, seed=0):
    from Unet import Unet
    from AutoencoderKL import AutoencoderKL
    unet, vae = Unet().eval(), AutoencoderKL().eval()
    fill_state_dict(unet.state_dict(), seed=seed, fast=True)
    fill_state_dict(vae.state_dict(), seed=seed + 1, fast=True)
    return unet, vae


def cpu_baseline(unet, vae, L, steps, threads, mode="sample"):
    """The oracle (fp32 torch restatement == reference arithmetic) on the host cores: one UNet step (CFG batch 2) and
    one decode at the benchmark resolution, extrapolated to steps x UNet + 1 decode."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sd_oracle as O
    torch.set_num_threads(threads)
    sd_u, sd_v = unet.state_dict(), vae.state_dict()
    rs = np.random.RandomState(1)
    f32 = lambda *s: torch.from_numpy(rs.randn(*s).astype(np.float32))
    x, ctx, t = f32(2, 9, L, L), f32(2, 77, 1024), torch.tensor([501, 501])
    with torch.no_grad():
        if mode == "decode":
            z = f32(1, 4, L, L)
            t0 = time.perf_counter(); O.vae_decode(sd_v, z); t_dec = time.perf_counter() - t0
            return {"value": 1.0 / t_dec, "unit": "images/s", "cores": threads, "kind": "port",
                    "sample": "1 decode (%.2f s) at %dx%d, fp32 oracle" % (t_dec, 8 * L, 8 * L), "decode_s": t_dec}
        t0 = time.perf_counter(); O.unet_forward(sd_u, x, t, ctx); t_unet = time.perf_counter() - t0
        z = f32(1, 4, L, L)
        t0 = time.perf_counter(); O.vae_decode(sd_v, z); t_dec = time.perf_counter() - t0
    per_image = steps * t_unet + t_dec
    return {"value": 1.0 / per_image, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "1 UNet step (CFG batch 2, %.2f s) + 1 decode (%.2f s) at %dx%d, fp32 oracle, extrapolated to "
                      "%d steps + decode" % (t_unet, t_dec, 8 * L, 8 * L, steps),
            "unet_step_s": t_unet, "decode_s": t_dec}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=768)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--scale", type=float, default=9.0)
    ap.add_argument("--mode", choices=["sample", "decode"], default="sample",
                    help="sample: the full inpainting job (BASELINE configs 2-4); decode: AutoencoderKL.decode of --batch "
                         "latents per step, uint8 images out (BASELINE config 5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--pmc-workload", action="store_true",
                    help="only run one eager UNet step + one decode (after a warm-up pass of both): the workload of the rocprofv3 --pmc passes behind roofline.traffic (profiles/README.md)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SDMI_BENCH_FORCE_DIST=1: a one-rank launch still goes through the RCCL group (rehearsal of the N > 1 control flow
    # on a one-GPU box: barriers, the MAX all-reduce of the time, the image all-gather, rank 0's roofline pass)
    use_dist = world > 1 or os.environ.get("SDMI_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    _lib.load()
    torch.set_grad_enabled(False)

    L = args.res // 8
    b = args.batch
    unet_cpu, vae_cpu = build_models(dev)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(unet_cpu, vae_cpu, L, args.ddim_steps, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))), args.mode)
    unet, vae = unet_cpu.to(dev), vae_cpu.to(dev)
    inp = make_inputs(b, L, args.ddim_steps, seed=100 + rank)
    g = lambda t: t.to(dev)
    cond, uncond, c_concat, x_T = g(inp["cond"]), g(inp["uncond"]), g(inp["c_concat"]), g(inp["x_T"])
    noises = [g(n) for n in inp["noises"]]
    sampler = Sampler(unet, vae)
    total = b * world

    if args.pmc_workload:
        xc = torch.randn((2 * b, 9, L, L), device=dev)
        tb = torch.full((2 * b,), 501, dtype=torch.long, device=dev)
        cc = unet.precompute_context(torch.cat([uncond, cond]))
        for _ in range(2):   # pass 1 warms up (weight pre-pack); pass 2 is the one profiles/pmc_traffic.py reads
            unet(xc, timesteps=tb, context=cc)
            vae.decode(x_T / 0.18215)
            torch.cuda.synchronize()
        return

    def one_step():
        if args.mode == "decode":
            img = sampler.decode_to_uint8(x_T, use_graph=not args.no_graph)
        else:
            _, img = sampler.sample(args.ddim_steps, cond, uncond, c_concat, x_T, noises, args.scale, use_graph=not args.no_graph)
        return all_gather_images(img, total)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    barrier()
    dt = time.perf_counter() - t0
    assert out.shape[0] == total and out.dtype == torch.uint8
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # UNet step latency (graph replay of one forward, CFG batch 2b) on its own
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    unet_ms = None
    if args.mode == "sample":
        xc = torch.randn((b, 9, L, L), device=dev).repeat(2, 1, 1, 1)   # CFG: both halves see the same latent and conditioning
        tb = torch.full((2 * b,), 501, dtype=torch.long, device=dev)
        cc = sampler.make_context(uncond, cond)
        tb = unet.precompute_time_embedding(tb)   # as the sampling loop runs it: the t-only MLPs are hoisted out of the step
        sampler._unet_step(xc, tb, cc, not args.no_graph, same_halves=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            sampler._unet_step(xc, tb, cc, not args.no_graph, same_halves=True)
        e1.record()
        torch.cuda.synchronize()
        unet_ms = e0.elapsed_time(e1) / 10
    sampler.decode_to_uint8(x_T, use_graph=not args.no_graph)
    torch.cuda.synchronize()
    e0.record()
    sampler.decode_to_uint8(x_T, use_graph=not args.no_graph)
    e1.record()
    torch.cuda.synchronize()
    dec_ms = e0.elapsed_time(e1)

    # FLOPs the job actually executes (the CFG halves share a prefix that runs once; algorithmic figure beside it): the
    # work_of() sum over the C-ABI launches of one eager UNet step in the sampler's own execution form, and of one decode
    def executed_gflop(fn):
        with _lib.KernelTimer(work_of) as kt:
            fn()
        torch.cuda.synchronize()
        return sum(w[0] for _, w, _, _ in kt.records) / 1e9
    exec_unet = executed_gflop(lambda: sampler._unet_step(xc, tb, cc, False, same_halves=True)) if args.mode == "sample" else 0.0
    exec_dec = executed_gflop(lambda: sampler.decode_to_uint8(x_T, use_graph=False))

    def measured_peaks():
        """MFMA issue rate (v_mfma_f32_32x32x16 only, two waves per SIMD) and streaming-copy bandwidth of THIS box."""
        st = torch.cuda.current_stream()
        sp = ctypes.c_void_p(st.cuda_stream)
        scratch = torch.zeros(16, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        blocks, iters = 256 * 2, 20000
        _lib.call("sdmi_probe_mfma", scratch.data_ptr(), blocks, 200, sp)
        e0.record(st); _lib.call("sdmi_probe_mfma", scratch.data_ptr(), blocks, iters, sp); e1.record(st)
        torch.cuda.synchronize()
        mfma = blocks * 4 * iters * 8 * (2.0 * 32 * 32 * 16) / (e0.elapsed_time(e1) * 1e-3) / 1e12
        n = 1 << 30
        src, dst = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
        src.zero_()
        _lib.call("sdmi_probe_copy", src.data_ptr(), dst.data_ptr(), n, sp)
        e0.record(st)
        for _ in range(5):
            _lib.call("sdmi_probe_copy", src.data_ptr(), dst.data_ptr(), n, sp)
        e1.record(st)
        torch.cuda.synchronize()
        hbm = 5 * 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
        return {"mfma_tflops": mfma, "hbm_copy_gbs": hbm}

    roof, kernels = None, None
    traffic_file = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"))
                         if os.path.exists(f)), None)
    pmc_traffic = json.load(open(traffic_file)) if traffic_file else {}

    def traffic_of(kernel):
        """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md
        HBM section), weighted like `achieved`: ddim_steps x the UNet-step launches + the decode launches."""
        ks = pmc_traffic.get("kernels", {})
        # (the profiler prints the GEMM kernels with their last template argument - the staging variant - which the
        # library's own kernel note leaves out)
        e = ks.get(kernel) or next((v for k, v in ks.items() if k.startswith(kernel[:-1] + ",")), None)
        if not e or pmc_traffic.get("resolution") != args.res:
            return None
        byt = args.ddim_steps * e["unet"]["bytes"] + e["decode"]["bytes"]
        n = args.ddim_steps * e["unet"]["launches"] + e["decode"]["launches"]
        return byt / n if n else None

    if rank == 0 and not args.no_roofline:
        # dominant kernel: every C-ABI launch of one eager UNet step + one decode bracketed by HIP events on its stream
        # eager launches cost the host ~25 us each, more than many kernels run: give the GPU a head start of spinning so
        # that the launches queue up and each event pair brackets the kernel alone, not the wait for its launch
        # The per-kernel figures are taken with the CFG batch [uncond | cond] as ONE pass on one stream: the sampling loop
        # runs the same kernels as two concurrent batch-1 passes (same bits), where a kernel's event pair would also
        # time whatever the other stream runs beside it.
        if args.mode == "sample":
            cc1 = unet.precompute_context(torch.cat([uncond, cond]).contiguous())
            unet(xc, timesteps=tb, context=cc1)   # warm-up of the batch-2 shapes
            torch.cuda.synchronize()
        with _lib.KernelTimer(work_of) as kt:
            if args.mode == "sample":
                torch.cuda._sleep(int(60e6))
                unet(xc, timesteps=tb, context=cc1)
            n_unet = len(kt.records)
            zdec = x_T / 0.18215
            torch.cuda.synchronize()
            torch.cuda._sleep(int(60e6))
            vae.decode(zdec)
        summ = kt.summary()
        # weight each entry point by its share of an image: 50 UNet steps + 1 decode
        tot = {}
        for i, (name, work, a, c) in enumerate(kt.records):
            wgt = args.ddim_steps if i < n_unet else 1
            d = tot.setdefault(name, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            d["ms"] += wgt * a.elapsed_time(c); d["flops"] += wgt * work[0]; d["bytes"] += wgt * work[1]; d["launches"] += wgt
        dom = max(tot, key=lambda k: tot[k]["ms"])
        d = tot[dom]
        # the roof that binds this kernel: whichever of (algorithmic flops / dense MFMA peak) and (algorithmic bytes / HBM
        # peak) is the longer time - the K = 320 GEMMs of the 96x96 level move 100 B per kflop and sit under the HBM roof
        t_mfma, t_hbm = d["flops"] / (PEAK_MFMA_TFLOPS * 1e12), d["bytes"] / (PEAK_HBM_GBS * 1e9)
        both = {"frac_mfma": t_mfma / (d["ms"] * 1e-3), "frac_hbm": t_hbm / (d["ms"] * 1e-3),
                "flop_per_byte": d["flops"] / d["bytes"] if d["bytes"] else None}
        if d["flops"] > 0 and t_mfma >= t_hbm:
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / PEAK_MFMA_TFLOPS, "traffic": traffic_of(dom), "avg_launch_ms": d["ms"] / d["launches"],
                    "launches_per_image": d["launches"], "share_of_image_time": d["ms"] / sum(v["ms"] for v in tot.values())}
        else:
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": ach / PEAK_HBM_GBS, "traffic": traffic_of(dom), "avg_launch_ms": d["ms"] / d["launches"],
                    "launches_per_image": d["launches"], "share_of_image_time": d["ms"] / sum(v["ms"] for v in tot.values())}
        roof.update(both)
        # the same fraction against what this box measures: `peak` / `frac` above stay the nominal ones
        roof["traffic_source"] = ("committed PMC passes, %s (not re-measured in this run)" % os.path.relpath(traffic_file, ROOT)
                                  if roof["traffic"] is not None else None)
        mp = measured_peaks()
        roof["peak_measured"] = mp["mfma_tflops"] if roof["bound"] == "mfma" else mp["hbm_copy_gbs"]
        roof["frac_of_measured"] = roof["achieved"] / roof["peak_measured"]
        roof["measured_peaks"] = mp
        kernels = {k: {"ms_per_image": round(v["ms"], 3), "launches": v["launches"],
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                       "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                   for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["ms"])}

    if rank == 0:
        if args.mode == "decode":
            gflop_img, gflop_exec = DECODE_GFLOP.get(L, 0.0), exec_dec / b
            workload = ("AutoencoderKL.decode %dx%d, batch %d per GPU, latents -> uint8 HWC images, random-init weights at "
                        "checkpoint shapes" % (args.res, args.res, b))
            metric = "images/sec (whole node) VAE decode %dx%d" % (args.res, args.res)
        else:
            gflop_img = args.ddim_steps * UNET_GFLOP.get(L, 0.0) + DECODE_GFLOP.get(L, 0.0)
            gflop_exec = (args.ddim_steps * exec_unet + exec_dec) / b
            workload = ("SD-2 inpainting %dx%d, batch %d per GPU, %d DDIM steps, CFG scale %g, UNet + fused CFG/DDIM update + "
                        "AutoencoderKL decode, random-init weights at checkpoint shapes" % (args.res, args.res, b, args.ddim_steps, args.scale))
            metric = "images/sec (whole node) %dx%d %d-step inpaint; UNet step latency ms" % (args.res, args.res, args.ddim_steps)
        line = {
            "metric": metric,
            "value": total * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if E.act_dtype() == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": workload, "mode": args.mode,
                       "resolution": args.res, "batch_per_gpu": b, "ddim_steps": args.ddim_steps, "guidance_scale": args.scale,
                       "parallelism": "images sharded dp%d, one all-gather of uint8 outputs" % world,
                       "hip_graph": not args.no_graph},
            "unet_step_ms": unet_ms, "decode_ms": dec_ms,
            # executed: what the launches of this run compute (shared CFG prefix once); algorithmic: SURVEY 8(a) per image
            "mfma_frac_end_to_end": (gflop_exec * b / 1e3) / (dt / args.steps) / PEAK_MFMA_TFLOPS,
            "mfma_frac_algorithmic": (gflop_img * b / 1e3) / (dt / args.steps) / PEAK_MFMA_TFLOPS if gflop_img else None,
            "gflop_per_image": {"executed": gflop_exec, "algorithmic": gflop_img or None},
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
            "kernels_note": "HIP-event pairs around every C-ABI launch of ONE eager batch-%d UNet step (x%d) + one decode on one "
                            "stream; a ranking, not an accounting of the graph-replayed two-stream loop (rocprofv3 window of "
                            "the same pass: profiles/r03_roofline_pass_kernel_stats.csv)" % (2 * b, args.ddim_steps),
        }
        print(json.dumps(line))
    if use_dist:
        dist.barrier()   # rank 0 runs the roofline pass after the timed region: keep the group alive until it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
