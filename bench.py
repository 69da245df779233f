#!/usr/bin/env python3
"""Headline benchmark: Mrays/s at 1920x1080x128 (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W

A *step* renders the config-2 frame twice -- once with the normal shader (mode 4) and once
with the Lambertian shader (mode 0), the two shaders config 2 names -- through
RenderLoop.forward, i.e. the fused HIP kernel k_render_fwd (camera -> 128 march iterations
-> distance -> tetrahedral normals -> shader).  Camera buffers, scene parameters and output
images are resident in HBM before the timed region; nothing crosses PCIe inside it.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling by pixel-row tiles.
The frame is 1920 x (1080*N); rank r renders rows [1080 r, 1080 (r+1)) and the tiles are
gathered to rank 0 over RCCL on a side stream, overlapped with the next frame's render.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PX = 3.45e-6
W, H_TILE, STEPS_MARCH, EPS = 1920, 1080, 128, 5e-2
MODES = (4, 0)                 # normal, lambertian
BYTES_PER_RAY = 36             # 12 B origin + 12 B direction + 12 B RGB
FLOPS_PER_EVAL = 80            # scene2, incl. 5 sqrt
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL tile gather")
    ap.add_argument("--no-early-out", action="store_true")
    ap.add_argument("--linear-waves", action="store_true", help="64x1 pixels per wave instead of 8x8 tiles")
    ap.add_argument("--static-tiles", action="store_true", help="static tile striding instead of the atomic tile queues")
    ap.add_argument("--precision", default="exact", choices=["exact", "fast"],
                    help="exact (default, headline): bit-faithful arithmetic; fast: opt-in 1-ulp sqrt + FMA contraction")
    ap.add_argument("--camera-z", type=float, default=-3.0, help="camera position (0,0,z); SURVEY 8d uses -3 and +1")
    ap.add_argument("--skip-backward", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the secondary 2-stream pipelined measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (single-GPU box)")
    ap.add_argument("--graph-leg", action="store_true", help=argparse.SUPPRESS)   # child process of the backward probe
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_cores():
    """Threads this process may really use: the affinity mask, capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a 16-CPU box share)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("RM_CPU_THREADS", "16"))))


def cpu_baseline():
    """The oracle (eager PyTorch CPU restatement of the reference, bit-exact with it in the build
    container) on this box's host cores, on the SAME workload as one benchmark step: the full
    1920x1080 ray grid, 128 steps, normal + Lambertian frames; one warm-up pass, best of 2
    (about 10 s of CPU work on 16 threads)."""
    from oracle import sdf_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} host threads ...")
    bufs = O.camera_buffers(1, W, H_TILE, PX * H_TILE, PX * W, PX * H_TILE)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]])
    t = torch.tensor([[0.0, 0.0, -3.0]])
    spec = O.scene_test2()
    rays = W * H_TILE * len(MODES)
    best = float("inf")
    with torch.no_grad():
        for it in range(3):
            t0 = time.perf_counter()
            for m in MODES:
                O.render(spec, bufs, q, t, m, 1, STEPS_MARCH, EPS)
            dt = time.perf_counter() - t0
            log(f"cpu_baseline: pass {it} {dt:.2f} s")
            if it > 0:
                best = min(best, dt)
    return {"value": rays / best / 1e6, "unit": "Mrays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "one full benchmark step: 1920x1080 rays x128 steps, normal+lambertian frames, "
                      "oracle/sdf_oracle.py eager fp32, best of 2 after a warm-up",
            "seconds": best}


def pipelined_probe(loop, q, t, rows, dev, rays_per_frame, frames=40, nstreams=2):
    """Secondary number, outside the timed region: the same frames issued round-robin on two HIP streams,
    so one frame's straggler tail overlaps the next frame's start (independent frames in flight, as a
    display loop would double-buffer).  The headline `value` stays the serial, one-stream measurement."""
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(n):
                with torch.cuda.stream(streams[i % nstreams]):
                    loop(q, t, MODES[i % len(MODES)], 1, STEPS_MARCH, rows=rows)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(6)
    dt = run(frames)
    return {"streams": nstreams, "frames": frames, "ms_per_frame": dt / frames * 1e3,
            "value": rays_per_frame * frames / dt / 1e6, "unit": "Mrays/s"}


def other_camera_probe(loop, q, rows, dev, rays_per_frame, timed_z, frames=20):
    """SURVEY 8(d) names two cameras for config 2: (0,0,-3), outside the torus (the timed one by default),
    and the reference's default pose (0,0,1) (main.py:46), inside the torus tube, where more rays never
    settle.  Report the one that was not timed, measured the same way (serial, one stream)."""
    z = 1.0 if timed_z != 1.0 else -3.0
    t = torch.tensor([[0.0, 0.0, z]], device=dev)
    with torch.no_grad():
        for i in range(4):
            loop(q, t, MODES[i % len(MODES)], 1, STEPS_MARCH, rows=rows)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            loop(q, t, MODES[i % len(MODES)], 1, STEPS_MARCH, rows=rows)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"camera": [0.0, 0.0, z], "ms_per_frame": dt / frames * 1e3, "value": rays_per_frame * frames / dt / 1e6,
            "unit": "Mrays/s"}


def backward_probe(dev):
    """Secondary metric 'fwd+bwd ms/frame' (BASELINE config 4 shape): closed make_test_scene,
    512x512, 64 steps, Lambertian MSE loss, gradients of all 40 scene parameters."""
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    h = w = 512
    scene = make_closed_test_scene()
    loop = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=PX * h, sensor_width=PX * w,
                      sensor_height=PX * h, normals_eps=EPS).to(dev)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
    t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
    target = torch.rand(1, h, w, 1, device=dev)
    def iteration(events=None):
        for p in scene.parameters():
            p.grad = None
        if events is not None:
            events[0].record()
        loss = (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean()
        if events is not None:
            events[1].record()
        loss.backward()
        if events is not None:
            events[2].record()

    # (1) latency of ONE step issued into an idle GPU and waited for (synchronise after every step)
    fwd = bwd = 0.0
    reps = 5
    for it in range(reps + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        iteration(ev)
        torch.cuda.synchronize()
        if it > 0:
            fwd += ev[0].elapsed_time(ev[1])
            bwd += ev[1].elapsed_time(ev[2])
    sync_fwd, sync_bwd = fwd / reps, bwd / reps
    # (2) a training loop as it is written in practice: steps issued back to back, nothing synchronises
    # inside (no timing events either: recorded inside this loop they slowed the host side of the forward
    # by 0.3 ms per step when the probe ran after the main benchmark, not when it ran alone)
    n = 100
    for _ in range(10):
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        iteration()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    out = {"config": "closed make_test_scene 512x512x64, lambertian MSE, 40 parameters",
           "fwd_bwd_ms": wall,
           "note": "fwd_bwd_ms = wall time per step of a 100-step eager loop with no synchronisation inside (GPU-bound: "
                   "the two kernels take 0.18 + 0.36 ms); *_sync_ms = one step issued into an idle GPU and waited for, "
                   "split by events; graph_fwd_bwd_ms = the same step replayed from a HIP graph",
           "fwd_sync_ms": sync_fwd, "bwd_sync_ms": sync_bwd, "fwd_bwd_sync_ms": sync_fwd + sync_bwd}
    # the same forward + backward captured once in a HIP graph (torch.cuda.graph) and replayed: the eager
    # figure above is mostly host time between ~25 small launches, the replay is GPU time.  Runs in a child
    # process: torch's capture of backward() has crashed here (under rocprofv3, and on parameters whose
    # AccumulateGrad nodes were made by earlier eager iterations), and that must not cost the bench line.
    out["graph_fwd_bwd_ms"] = None
    if not any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")):
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-leg"], capture_output=True,
                               text=True, timeout=180)
            if r.returncode == 0:
                out["graph_fwd_bwd_ms"] = json.loads(r.stdout.strip().splitlines()[-1])["graph_fwd_bwd_ms"]
            else:
                log(f"graph leg exited with {r.returncode}")
        except Exception as e:       # noqa: BLE001  (timeout, bad output: report null)
            log(f"graph leg failed: {e}")
    return out


def graph_leg():
    """Child process of backward_probe: forward + backward of the config-4 shape captured with
    torch.cuda.graph and replayed; prints {"graph_fwd_bwd_ms": ...}."""
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    h = w = 512
    scene = make_closed_test_scene()
    loop = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=PX * h, sensor_width=PX * w,
                      sensor_height=PX * h, normals_eps=EPS).to(dev)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
    t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
    target = torch.rand(1, h, w, 1, device=dev)
    params = list(scene.parameters())

    def step():
        (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean().backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            for p in params:
                p.grad = None
            step()
    torch.cuda.current_stream().wait_stream(side)
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    print(json.dumps({"graph_fwd_bwd_ms": (time.perf_counter() - t0) / n * 1e3}), flush=True)


def main():
    args = parse()
    if args.graph_leg:
        return graph_leg()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build_library()
        if os.environ.get("RM_SPECIALIZE", "auto") != "off":
            # the per-scene libraries of the two benchmark scenes (no-op when __graft_entry__.build() made
            # them; ~10 s of hipcc each on a fresh checkout, where the "auto" policy would otherwise time
            # the interpreter while the library builds in the background)
            from ray_marching_amd import specialize
            from ray_marching_amd.compiler import compile_scene
            from ray_marching_amd.scene.scene_registry import make_closed_test_scene, make_test_scene2 as _s2
            try:
                specialize.build(compile_scene(_s2()), precision=args.precision)
                if not args.skip_backward:
                    specialize.build(compile_scene(make_closed_test_scene()))
            except Exception as e:      # noqa: BLE001  (no hipcc: the interpreter still renders the frame)
                log(f"specialised libraries not built: {e}")
    if dist is not None:
        dist.barrier()

    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene.scene_registry import make_test_scene2

    h_total = H_TILE * world
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=h_total, focal_length=PX * H_TILE,
                      sensor_width=PX * W, sensor_height=PX * h_total, normals_eps=EPS,
                      early_out=not args.no_early_out, tile8x8=not args.linear_waves, dynamic_tiles=not args.static_tiles, precision=args.precision).to(dev)
    from ray_marching_amd.compiler import compiled_for
    specialised = compiled_for(loop.scene).specialised
    rows = (rank * H_TILE, (rank + 1) * H_TILE)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
    t = torch.tensor([[0.0, 0.0, args.camera_z]], device=dev)
    rays_per_frame = W * H_TILE

    gather = dist is not None and not args.no_gather
    comm = torch.cuda.Stream(device=dev) if gather else None
    gathered = None
    if gather and rank == 0:
        # the Lambertian frame is one value in three channels: it crosses xGMI as one channel (distributed.py)
        from ray_marching_amd.distributed import GREY_MODES
        gathered = [[torch.empty(1, H_TILE, W, 1 if m in GREY_MODES else 3, device=dev) for _ in range(world)]
                    for m in MODES]

    kernel_ms = []

    def one_step(step_idx, timed):
        handles = []
        from ray_marching_amd import ops
        from ray_marching_amd.distributed import tile_payload
        for j, mode in enumerate(MODES):
            # HIP events on the launch stream immediately around the k_render_fwd launch (ops.Render.run)
            ops.kernel_event_sink = kernel_ms if timed else None
            img = loop(q, t, mode, 1, STEPS_MARCH, rows=rows)
            ops.kernel_event_sink = None
            if gather:
                done = torch.cuda.Event()
                done.record()
                with torch.cuda.stream(comm):
                    comm.wait_event(done)
                    img.record_stream(comm)
                    pay = tile_payload(img, mode)
                    if args.backend == "nccl":
                        dist.gather(pay, gathered[j] if rank == 0 else None, dst=0)
                    else:   # rehearsal backends move host tensors
                        host = pay.cpu()
                        parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                        dist.gather(host, parts, dst=0)
            handles.append(img)
        return handles

    log(f"rank {rank}/{world}: warm-up {args.warmup}, timing {args.steps} steps")
    with torch.no_grad():
        for i in range(args.warmup):
            one_step(i, False)
        if comm is not None:
            torch.cuda.current_stream().wait_stream(comm)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(i, True)
        if comm is not None:
            torch.cuda.current_stream().wait_stream(comm)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0

    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    log(f"timed region {elapsed:.3f} s")
    per_launch_ms = sum(a.elapsed_time(b) for a, b in kernel_ms) / max(len(kernel_ms), 1)
    total_rays = rays_per_frame * len(MODES) * args.steps * world
    value = total_rays / elapsed / 1e6

    if rank == 0:
        evals = rays_per_frame * (STEPS_MARCH + 6)
        ach_gbs = rays_per_frame * BYTES_PER_RAY / (per_launch_ms * 1e-3) / 1e9
        ach_tf = evals * FLOPS_PER_EVAL / (per_launch_ms * 1e-3) / 1e12
        traffic = valu_insts = None
        prof = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(prof):
            with open(prof) as f:
                pmc = json.load(f)
            traffic = pmc.get("k_render_fwd_hbm_bytes_per_launch")
            if specialised and args.precision == "exact" and not args.no_early_out and not args.linear_waves:
                valu_insts = pmc.get("k_render_fwd_valu_wave_instructions_per_launch")
        out = {
            "metric": "Mrays/sec at 1920x1080x128 iters; fwd+bwd ms/frame",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: make_test_scene2 (room shell + sphere/torus/capsule), "
                                   "1920x1080 pinhole ray grid per GPU, 128 march iters, normal + Lambertian "
                                   "frames per step, fp32", "frames_per_step": len(MODES),
                       "rays_per_frame": rays_per_frame, "camera": [0.0, 0.0, args.camera_z],
                       "early_out": not args.no_early_out, "wave_tile": "64x1" if args.linear_waves else "8x8", "arithmetic": args.precision,
                       "tile_schedule": "static stride" if args.static_tiles else "64 atomic queues + stealing",
                       "kernels": "per-scene specialised (StaticCfg)" if specialised else "generic LDS interpreter",
                       "parallelism": f"row-tiles x{world}" + (" + RCCL gather" if gather else "")},
            "ray_sdf_evals_per_s": value * 1e6 * (STEPS_MARCH + 6),
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_render_fwd", "kernel_ms": per_launch_ms,
                         "algorithmic_bytes_per_launch": rays_per_frame * BYTES_PER_RAY,
                         "binding": "fp32-valu (SURVEY D8: 298 flop/B, the fused frame cannot be HBM bound)",
                         "valu": {"achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": ach_tf / VALU_PEAK_TFLOPS,
                                  "algorithmic_flops_per_launch": evals * FLOPS_PER_EVAL,
                                  "note": "algorithmic = R*(S+6)*80 flop; the bit-exact early-out executes fewer",
                                  # executed instructions (SQ_INSTS_VALU, PMC pass in profiles/) against this
                                  # launch time: SIMD cycles per VALU wave-instruction.  2 = the fp32 rate of
                                  # two-VGPR-operand arithmetic; ~4 is what this mix of compares, selects,
                                  # min/max, 3-operand FMAs and scalar operands can issue (DESIGN.md section 8,
                                  # profiles/micro/valu_issue_bench.hip)
                                  "executed_wave_instructions": valu_insts,
                                  "simd_cycles_per_instruction": (None if not valu_insts else
                                                                  per_launch_ms * 1e-3 * 2.4e9 * 1024 / valu_insts),
                                  "modelled_issue_cycles_per_instruction": 4.0}},
        }
        if world == 1 and not args.no_pipelined:
            out["other_camera"] = other_camera_probe(loop, q, rows, dev, rays_per_frame, args.camera_z)
            out["pipelined"] = pipelined_probe(loop, q, t, rows, dev, rays_per_frame)
        if not args.skip_backward:
            log("backward probe (config 4 shape) ...")
            out["fwd_bwd"] = backward_probe(dev)
            log(f"backward probe: {out['fwd_bwd']}")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
