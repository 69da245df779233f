#!/usr/bin/env python3
"""Headline benchmark: Mrays/s at 1920x1080x128 (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W

A *step* renders the config-2 frame twice -- once with the normal shader (mode 4) and once
with the Lambertian shader (mode 0), the two shaders config 2 names -- through
RenderLoop.forward, i.e. the fused HIP kernel k_render_fwd (camera -> 128 march iterations
-> distance -> tetrahedral normals -> shader).  Camera buffers, scene parameters and output
images are resident in HBM before the timed region; nothing crosses PCIe inside it.

N > 1, one rank per GPU over RCCL: started by torch.distributed.run (RANK / WORLD_SIZE in the environment), or --
a plain `python bench.py --gpus N` -- by this script itself, which starts the N ranks as fresh child processes
BEFORE it makes any GPU call, relays rank 0's JSON line and exits non-zero if a rank fails or the node has fewer
than N GPUs.  Weak scaling: the job is an N-camera batch of the config-2 frame (the reference's `num_cameras`,
control.py:201), one 1920x1080 camera per rank at the same pose, so every rank does EQUAL work; the frames reach
rank 0 as point-to-point transfers (one xGMI link per peer, all in flight together) on a side stream, overlapped with
the next frame's render.  `--weak-mode tall` is the round-2 variant: one 1920 x (1080*N) frame at fixed focal length,
rank r renders rows [1080 r, 1080 (r+1)) -- bands see different content, i.e. unequal work per rank.

`--config 5` times BASELINE configs[4] instead: the 7680x4320 frame of the 32-primitive smooth-union scene at 256
steps, STRONG scaling over row bands (N = 1 renders all 8 bands itself).

The timed region is repeated (`--repeats`, default 5 blocks of K steps, each bracketed by barrier + synchronise);
`value` is the median block, the spread is reported next to it.

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
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; value = their median")
    ap.add_argument("--config", type=int, default=2, choices=[2, 5], help="2: headline (configs[1]); 5: configs[4], strong scaling")
    ap.add_argument("--exchange", default="p2p", choices=["p2p", "gather"], help="N>1 tile exchange: batch_isend_irecv or dist.gather")
    ap.add_argument("--skip-config3", action="store_true", help="skip the secondary 3840x2160x256 fp16 measurement")
    ap.add_argument("--weak-mode", default="cameras", choices=["cameras", "tall"],
                    help="N>1, config 2: one camera of an N-camera batch per rank (equal work, default) or row bands of one tall frame")
    ap.add_argument("--stub-render", action="store_true",
                    help="launcher / collective rehearsal without a GPU: a host stand-in replaces the HIP RenderLoop; the line is marked stub")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)   # tests: this rank raises before the first barrier
    ap.add_argument("--graph-leg", action="store_true", help=argparse.SUPPRESS)   # child process of the backward probe
    ap.add_argument("--interp-leg", action="store_true", help=argparse.SUPPRESS)  # child process: the frame through the LDS interpreter
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
    run(40)         # per stream: workspace batches, the regen="auto" probe of the first cycle and its decision
    dt = run(frames)
    return {"streams": nstreams, "frames": frames, "ms_per_frame": dt / frames * 1e3,
            "value": rays_per_frame * frames / dt / 1e6, "unit": "Mrays/s"}


def kernel_in_use(loop):
    """Which frame kernel RenderLoop(regen='auto') settled on for the launches timed last (None: not applicable)."""
    if loop.regen != "auto":
        return "k_march_regen + k_render_finish" if loop.regen else "k_render_fwd"
    states = list(loop._choice_state.values())
    if not states:
        return "k_render_fwd"
    return "k_march_regen + k_render_finish" if states[-1]["regen"] else "k_render_fwd"


def frame_rate(loop, q, t, rows, frames=20, modes=MODES, steps=STEPS_MARCH, warm=4, blocks=3):
    """Serial, one-stream frame time of `loop` at pose (q, t): ms per frame, median of `blocks` timed blocks (a single
    block now and then contains a host stall of tens of milliseconds on a shared box)."""
    times = []
    with torch.no_grad():
        for i in range(warm):
            loop(q, t, modes[i % len(modes)], 1, steps, rows=rows)
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(frames):
                loop(q, t, modes[i % len(modes)], 1, steps, rows=rows)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / frames * 1e3)
    return sorted(times)[len(times) // 2]


def config3_probe(dev):
    """BASELINE configs[2]: make_test_scene2 at 3840x2160, 256 march steps, module .to(float16) (the reference's
    own run dtype, main.py:20-26).  The kernel reads fp16 camera buffers / pose / parameters and writes fp16 RGB:
    18 B/ray of algorithmic HBM traffic, no cast passes (tests/test_gpu_fullsize.py)."""
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    h, w, steps = 2160, 3840, 256
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=PX * h,
                      sensor_width=PX * w, sensor_height=PX * h, normals_eps=EPS).to(dev).to(torch.float16)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev, dtype=torch.float16)
    out = {"config": "BASELINE configs[2]: make_test_scene2 3840x2160x256, float16 I/O (fp32 arithmetic), normal + lambertian frames",
           "bytes_per_ray": 18}
    for z in (-3.0, 1.0):
        t = torch.tensor([[0.0, 0.0, z]], device=dev, dtype=torch.float16)
        ms = frame_rate(loop, q, t, None, frames=16, steps=steps, warm=36)
        out[f"camera_z{z:+g}"] = {"ms_per_frame": ms, "value": h * w / ms / 1e3, "unit": "Mrays/s",
                                  "hbm_frac": h * w * 18 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": kernel_in_use(loop)}
        loop._choice_state.clear()
    return out


def reference_shape_probe(dev):
    """The reference's own run shape (main.py:20-26, 46, 65-88): make_test_scene2, 1440x900, 32 march steps, the
    module cast with .to(float16), pose (0,0,1), one frame = render_loop(...) followed by the display contract
    F.pad(images.mean(0).float(), [0,1], 1.0) -> [H,W,4] fp32 (window.draw's input, torchwindow/window.py:146-174).
    `fps_waited` waits for every frame like main.py's loop does (window.draw copies on the legacy stream);
    `ms_per_frame` is the same loop without the wait.  Modes: lambertian (main.py's first mode) and normal."""
    import torch.nn.functional as F
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    h, w, steps = 900, 1440, 32
    loop = RenderLoop(make_test_scene2().to(dev, torch.float16), num_cameras=1, px_width=w, px_height=h, focal_length=PX * h,
                      sensor_width=PX * w, sensor_height=PX * h, normals_eps=EPS).to(dev, torch.float16)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev, dtype=torch.float16)
    t = torch.tensor([[0.0, 0.0, 1.0]], device=dev, dtype=torch.float16)
    out = {"shape": "make_test_scene2 1440x900, 32 steps, float16 module, pose (0,0,1) (main.py:20-26,46)", "modes": {}}

    def frame(mode):
        images = loop(q, t, mode, 1, steps)
        return F.pad(images.mean(dim=0).float(), pad=[0, 1], value=1.0)

    with torch.no_grad():
        for mode, name in ((0, "lambertian"), (4, "normal")):
            for _ in range(20):
                frame(mode)
            torch.cuda.synchronize()
            n = 200
            t0 = time.perf_counter()
            for _ in range(n):
                frame(mode)
            torch.cuda.synchronize()
            free = (time.perf_counter() - t0) / n * 1e3
            t0 = time.perf_counter()
            for _ in range(n):
                frame(mode)
                torch.cuda.synchronize()
            waited = (time.perf_counter() - t0) / n * 1e3
            t0 = time.perf_counter()
            for _ in range(n):
                loop(q, t, mode, 1, steps)
            torch.cuda.synchronize()
            render = (time.perf_counter() - t0) / n * 1e3
            # RenderLoop.display_frame: the frame kernel stores the [H,W,4] fp32 display tensor itself (one launch)
            for _ in range(10):
                loop.display_frame(q, t, mode, 1, steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                loop.display_frame(q, t, mode, 1, steps)
                torch.cuda.synchronize()
            fused_waited = (time.perf_counter() - t0) / n * 1e3
            # ... and replayed from a HIP graph (RenderLoop.capture(display=True)): the host side of a 0.14 ms frame matters
            shot = loop.capture(mode, 1, steps, display=True)
            for _ in range(10):
                shot(q, t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                shot(q, t)
                torch.cuda.synchronize()
            graph_waited = (time.perf_counter() - t0) / n * 1e3
            del shot
            out["modes"][name] = {"ms_per_frame": free, "fps": 1e3 / free, "ms_per_frame_waited": waited,
                                  "fps_waited": 1e3 / waited, "render_only_ms": render,
                                  "display_frame_ms_waited": fused_waited, "display_frame_fps_waited": 1e3 / fused_waited,
                                  "display_graph_ms_waited": graph_waited, "display_graph_fps_waited": 1e3 / graph_waited,
                                  "Mrays_per_s": h * w / free / 1e3}
    return out


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
    walls = []
    for _ in range(5):              # the eager loop is as much host- as GPU-bound (0.3-0.4 ms of Python per step): median of 5
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            iteration()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / n * 1e3)
    wall = sorted(walls)[len(walls) // 2]
    # (3) the same loop through RenderLoop.training_step: forward + loss + backward captured into one HIP graph on first
    # use, replayed per iteration (pose copied into static buffers): the training loop off the host
    helper_ms = None
    if not any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")):
        for p in scene.parameters():
            p.grad = None
        stepper = loop.training_step(lambda image: (image[..., :1] - target).pow(2).mean(), mode=0, marching_steps=64)
        for _ in range(5):
            stepper(q, t)
        runs = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                stepper(q, t)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / n * 1e3)
        helper_ms = sorted(runs)[len(runs) // 2]
        del stepper
    rays, S = h * w, 64
    n_bytes = step_bytes(rays, S)
    traffic, why = backward_traffic_record(h)
    if why:
        log(f"fwd_bwd.roofline.traffic: {why}")
    out = {"config": "closed make_test_scene 512x512x64, lambertian MSE, 40 parameters",
           "fwd_bwd_ms": wall,
           "training_step_ms": helper_ms,
           "note": "fwd_bwd_ms = wall time per step of a 100-step eager loop with no synchronisation inside; training_step_ms = "
                   "the same loop through RenderLoop.training_step (one HIP graph per iteration, captured on first use); *_sync_ms = "
                   "one step issued into an idle GPU and waited for, split by events; graph_fwd_bwd_ms = the same "
                   "step replayed from a HIP graph.  Kernels of a step: k_render_fwd (recording), k_render_bwd, "
                   "k_bwd_hard_n/_a/_b (rays whose march did not settle, evaluated per (ray, step) in parallel), "
                   "reductions; per-kernel times in profiles/r03_bwd_kernel_stats_512.csv",
           "fwd_bwd_ms_runs": walls,
           "fwd_sync_ms": sync_fwd, "bwd_sync_ms": sync_bwd, "fwd_bwd_sync_ms": sync_fwd + sync_bwd,
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "algorithmic_bytes_per_step": n_bytes,
                        "achieved": n_bytes / (wall * 1e-3) / 1e9, "frac": n_bytes / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic,
                        "binding": "neither roof: at this frame size the step is bound by VALU issue at 2-4 waves per SIMD "
                                   "and by the longest wave tile of each launch (DESIGN.md 7)"}}
    # the same forward + backward captured once in a HIP graph (torch.cuda.graph) and replayed: the eager
    # figure above includes host time between ~30 small launches, the replay is GPU time.  The capture recipe is
    # ray_marching_amd/graphs.py (warm-up and capture on one stream: round 1's aborted captures came from
    # AccumulateGrad nodes that belonged to the warm-up stream).  Still run in a child process, as isolation
    # only: whatever a capture does, it must not cost the bench line.
    out["graph_fwd_bwd_ms"] = None
    if not any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")):
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-leg"], capture_output=True,
                               text=True, timeout=300)
            if r.returncode == 0:
                leg = json.loads(r.stdout.strip().splitlines()[-1])
                out["graph_fwd_bwd_ms"] = leg["graph_fwd_bwd_ms"]
                # throughput mode: the same step at 1024^2 and 2048^2 (512^2 is 4 wave tiles per SIMD: a latency floor)
                out["size_sweep"] = leg["sweep"]
                g = n_bytes / (leg["graph_fwd_bwd_ms"] * 1e-3) / 1e9
                out["roofline"]["graph"] = {"achieved": g, "frac": g / HBM_PEAK_GBS}
                # the replay is the GPU work of a step; an eager loop well above it is waiting for the host (Python,
                # autograd's 40 AccumulateGrad nodes, ~20 launches), not for the kernels
                out["eager_loop_is_host_bound"] = bool(wall > 1.1 * out["graph_fwd_bwd_ms"])
            else:
                log(f"graph leg exited with {r.returncode}")
        except Exception as e:       # noqa: BLE001  (timeout, bad output: report null)
            log(f"graph leg failed: {e}")
    return out


def step_bytes(rays, S=64):
    """Algorithmic HBM bytes of one config-4 training step: forward 24 in + 12 out + 12 p_final + the
    trajectory written (12 S); backward reads the trajectory, p_final, the image gradient and the ray directions
    (12 (S + 3))."""
    return rays * ((24 + 12 + 12 + 12 * S) + 12 * (S + 3))


def graph_leg():
    """Child process of backward_probe: forward + backward of the config-4 scene captured with torch.cuda.graph and
    replayed, at 512^2 (BASELINE configs[3]) and, as a size sweep, at 1024^2 and 2048^2 -- what the kernels reach
    once the GPU is full instead of 4 wave tiles per SIMD.  Prints {"graph_fwd_bwd_ms": ..., "sweep": {...}}."""
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.graphs import capture_step
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
    t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
    out = {"sweep": {}}
    for size in (512, 1024, 2048):
        h = w = size
        scene = make_closed_test_scene()
        loop = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=PX * h, sensor_width=PX * w,
                          sensor_height=PX * h, normals_eps=EPS).to(dev)
        target = torch.rand(1, h, w, 1, device=dev)
        params = list(scene.parameters())

        def step():
            (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean().backward()

        graph, _, _ = capture_step(step, params, warmup=3)      # warm-up and capture on ONE stream (graphs.py)
        graph.replay()
        torch.cuda.synchronize()
        n = 50 if size == 512 else 12
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(n):
                graph.replay()
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / n * 1e3)
        ms = sorted(runs)[1]
        gbs = step_bytes(h * w) / (ms * 1e-3) / 1e9
        out["sweep"][f"{size}x{size}"] = {"graph_fwd_bwd_ms": ms, "achieved_GBps": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                                            "Mrays_per_s": h * w / ms / 1e3}
        if size == 512:
            out["graph_fwd_bwd_ms"] = ms
        del graph, loop, scene, target, params
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


def backward_traffic_record(size):
    """HBM bytes of one config-4 training step from the PMC passes (profiles/collect_r03.sh ->
    profiles/traffic_bwd.json) -- only if measured on THESE kernel sources."""
    prof = os.path.join(ROOT, "profiles", "traffic_bwd.json")
    if not os.path.isfile(prof):
        return None, "profiles/traffic_bwd.json missing"
    with open(prof) as f:
        pmc = json.load(f)
    from ray_marching_amd import _build
    if pmc.get("sources_hash") != _build.sources_hash():
        return None, "profiles/traffic_bwd.json was measured on other kernel sources (hash mismatch): traffic = null"
    return pmc.get(f"hbm_bytes_per_step_{size}"), None


def interp_leg():
    """Child process (RM_SPECIALIZE=off): the config-2 frame through the generic LDS interpreter -- what a scene the
    library has no specialised kernels for renders at until its background build has finished -- and how long that
    build takes on this box's host (hipcc of the per-scene library into a scratch directory)."""
    import tempfile
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H_TILE, focal_length=PX * H_TILE,
                      sensor_width=PX * W, sensor_height=PX * H_TILE, normals_eps=EPS).to(dev)
    assert not compiled_for(loop.scene).specialised
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
    t = torch.tensor([[0.0, 0.0, -3.0]], device=dev)
    ms = frame_rate(loop, q, t, None, frames=10, warm=3)
    out = {"ms_per_frame": ms, "value": W * H_TILE / ms / 1e3, "unit": "Mrays/s", "build_s": None}
    try:
        from ray_marching_amd import specialize
        specialize.SPEC_DIR = tempfile.mkdtemp(prefix="rm_spec_probe_")
        t0 = time.perf_counter()
        specialize.build(compiled_for(loop.scene), force=True)
        out["build_s"] = time.perf_counter() - t0
    except Exception as e:      # noqa: BLE001  (no hipcc on this box)
        log(f"interp leg: build not timed: {e}")
    print(json.dumps(out), flush=True)


def interpreter_record():
    """`first_frames` field: see interp_leg."""
    import subprocess
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--interp-leg"], capture_output=True, text=True,
                           timeout=240, env=dict(os.environ, RM_SPECIALIZE="off"))
        if r.returncode != 0:
            log(f"interp leg exited with {r.returncode}: {r.stderr[-300:]}")
            return None
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        rec["note"] = ("config-2 frame through the generic LDS interpreter (RM_SPECIALIZE=off): what an unknown scene renders at "
                       "until its per-scene library is built (build_s of hipcc, in a background thread, from the 64th frame on; "
                       "specialize.ensure(scene) builds it up front)")
        return rec
    except Exception as e:      # noqa: BLE001
        log(f"interp leg failed: {e}")
        return None


def traffic_record(specialised, args):
    """HBM traffic / executed-instruction counters of k_render_fwd from the PMC passes (profiles/collect.sh ->
    profiles/traffic.json) -- only if they were measured on THESE kernel sources and this configuration."""
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.isfile(prof):
        return None, None, "profiles/traffic.json missing"
    with open(prof) as f:
        pmc = json.load(f)
    from ray_marching_amd import _build
    if pmc.get("sources_hash") != _build.sources_hash():
        return None, None, "profiles/traffic.json was measured on other kernel sources (hash mismatch): traffic = null"
    if not (specialised and args.precision == "exact" and not args.no_early_out and not args.linear_waves
            and not args.static_tiles and args.camera_z == -3.0):
        return None, None, "non-default kernel configuration: the PMC record does not apply"
    return pmc.get("k_render_fwd_hbm_bytes_per_launch"), pmc.get("k_render_fwd_valu_wave_instructions_per_launch"), None


def exchange_tile(dist, args, pay, gathered_j, rank, world, device):
    """One tile to rank 0: point-to-point (a receive per peer on the root, posted together) or dist.gather."""
    if args.backend != "nccl":            # rehearsal backends move host tensors
        host = pay.cpu()
        if args.exchange == "gather":
            parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            return
        if rank == 0:
            bufs = [torch.empty_like(host) for _ in range(world - 1)]
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, r + 1) for r, b in enumerate(bufs)])
        else:
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, host, 0)])
        for r in reqs:
            r.wait()
        return
    if args.exchange == "gather":
        dist.gather(pay, gathered_j if rank == 0 else None, dst=0)
        return
    if rank == 0:
        gathered_j[0].copy_(pay)
        ops_ = [dist.P2POp(dist.irecv, gathered_j[r], r) for r in range(1, world)]
    else:
        ops_ = [dist.P2POp(dist.isend, pay, 0)]
    for r in dist.batch_isend_irecv(ops_):
        r.wait()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (torch.distributed.run,
    rendezvous on 127.0.0.1) before this process has made any GPU call -- never by replacing a process that has
    touched the GPU --, relay rank 0's JSON line, return non-zero when a rank failed or no line came back."""
    import socket
    import subprocess
    n = args.gpus
    if not (args.share_gpu or args.stub_render):
        have = torch.cuda.device_count()            # counts devices without initialising the GPU
        if have < n:
            log(f"--gpus {n} but this node shows {have} GPU(s); use --share-gpu only for single-GPU rehearsals")
            return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's peer mappings need it on this image
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    log(f"self-launch: {n} ranks via torch.distributed.run on 127.0.0.1:{port}")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)     # stderr passes through
    line = None
    for raw in proc.stdout:
        raw = raw.strip()
        if raw.startswith("{") and '"metric"' in raw:
            line = raw
        elif raw:
            print(raw, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        log(f"self-launch: torch.distributed.run exited with {rc} (a rank failed)")
        return rc
    if line is None:
        log("self-launch: the ranks finished without a result line")
        return 3
    print(line, flush=True)
    return 0


class StubLoop:
    """--stub-render: host stand-in for the HIP RenderLoop, so that the launcher, the process group, the barriers,
    the tile exchange and the JSON line can be rehearsed (and tested) on a machine without a GPU.  It renders nothing;
    the result line says so."""
    regen, _choice_state = False, {}

    def __init__(self, height, width):
        self.h, self.w = height, width

    def __call__(self, q, t, mode, degree, steps, rows=None):
        r0, r1 = rows if rows is not None else (0, self.h)
        return torch.full((1, r1 - r0, self.w, 3), float(mode))


def main():
    args = parse()
    if args.graph_leg:
        return graph_leg()
    if args.interp_leg:
        return interp_leg()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # before anything here touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    stub = args.stub_render
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        if not args.share_gpu and torch.cuda.device_count() < world:
            raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPU(s) visible (one rank per GPU; --share-gpu rehearses on one)")
        dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
        torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rank == args.stub_fail_rank:
            raise SystemExit(f"rank {rank}: failing on request (--stub-fail-rank)")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        world = dist.get_world_size()          # n_gpus of the line = what the backend really initialised

    def sync():
        if not stub:
            torch.cuda.synchronize()

    config5 = args.config == 5
    from ray_marching_amd.distributed import GREY_MODES, row_band, tile_payload
    if not stub:
        from ray_marching_amd.scene.scene_registry import make_closed_test_scene, make_many_primitive_scene, make_test_scene2
        make_scene = (lambda: make_many_primitive_scene(32)) if config5 else make_test_scene2

        import __graft_entry__ as entry
        if rank == 0:
            entry.build_library()
            if os.environ.get("RM_SPECIALIZE", "auto") != "off":
                # the per-scene libraries of the benchmark scenes (no-op when __graft_entry__.build() made them;
                # ~10 s of hipcc each on a fresh checkout, where the "auto" policy would otherwise time the
                # interpreter while the library builds in the background).  Built on rank 0 only, behind a barrier.
                from ray_marching_amd import specialize
                from ray_marching_amd.compiler import compile_scene
                try:
                    specialize.build(compile_scene(make_scene()), precision=args.precision)
                    if not args.skip_backward and not config5:
                        specialize.build(compile_scene(make_closed_test_scene()))
                except Exception as e:      # noqa: BLE001  (no hipcc: the interpreter still renders the frame)
                    log(f"specialised libraries not built: {e}")
    if dist is not None:
        dist.barrier()

    tile_w, tile_h = (W, H_TILE) if not stub else (96, 54)
    if config5:       # BASELINE configs[4]: one 7680x4320 frame, strong scaling over row bands
        width, h_total, march, modes, cam_z = (7680, 4320, 256, (4,), -4.5) if not stub else (96, 54, 256, (4,), -4.5)
        band = row_band(h_total, rank, world)
        focal, sensor_h = PX * h_total, PX * h_total
        work_per_rank = "row bands of one frame: unequal (the bands see different parts of the scene)" if world > 1 else "whole frame"
    elif args.weak_mode == "tall" and world > 1:
        # round-2 weak scaling: ONE frame of 1080*N rows at fixed focal length; the bands see different content
        width, h_total, march, modes, cam_z = tile_w, tile_h * world, STEPS_MARCH, MODES, args.camera_z
        band = (rank * tile_h, (rank + 1) * tile_h)
        focal, sensor_h = PX * tile_h, PX * h_total
        work_per_rank = "unequal (row bands of one taller frame at fixed focal length: top/bottom bands are mostly ceiling/floor)"
    else:
        # BASELINE configs[1] per GPU, weak scaling over a camera batch (the reference's num_cameras): every rank
        # renders one whole 1920x1080 camera at the same pose -> equal work; rank 0 collects the [N,H,W,C] batch
        width, h_total, march, modes, cam_z = tile_w, tile_h, STEPS_MARCH, MODES, args.camera_z
        band = (0, tile_h)
        focal, sensor_h = PX * tile_h, PX * tile_h
        work_per_rank = "equal (one 1920x1080 camera of an N-camera batch per rank, same pose)"
    banded = world > 1 and band != (0, h_total)          # the rank holds only ITS band of the camera buffers

    if stub:
        def make_loop(**kw):
            return StubLoop(h_total, width)
        specialised = False
    else:
        from ray_marching_amd.control import RenderLoop
        from ray_marching_amd.compiler import compiled_for

        def make_loop(**kw):
            return RenderLoop(make_scene(), num_cameras=1, px_width=width, px_height=h_total, focal_length=focal,
                              sensor_width=PX * width, sensor_height=sensor_h, normals_eps=EPS,
                              early_out=not args.no_early_out, tile8x8=not args.linear_waves,
                              dynamic_tiles=not args.static_tiles, precision=args.precision,
                              rows=band if banded else None, **kw).to(dev)
    loop = make_loop()
    if not stub:
        specialised = compiled_for(loop.scene).specialised
    rows = band if (stub and banded) else None          # a band loop renders its band by default
    band_rows = band[1] - band[0]
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
    t = torch.tensor([[0.0, 0.0, cam_z]], device=dev)
    rays_per_frame = width * (tile_h if not config5 else h_total)      # per GPU (weak) / whole frame (strong)

    gather = dist is not None and not args.no_gather
    comm = torch.cuda.Stream(device=dev) if (gather and not stub) else None
    gathered = None
    per = -(-h_total // world) if config5 else tile_h
    if gather and rank == 0:
        # ONE preallocated buffer per frame kind; rank r's tile is received straight into rows [r*per, (r+1)*per)
        # (contiguous views: no concatenation pass).  Grey shaders are one value in three channels: they cross xGMI as
        # one channel (distributed.py)
        gathered = []
        for m in modes:
            whole = torch.empty(1, per * world, width, 1 if m in GREY_MODES else 3, device=dev)
            gathered.append([whole[:, r * per:(r + 1) * per] for r in range(world)])

    kernel_ms = []

    def one_step(timed):
        if not stub:
            from ray_marching_amd import ops
        for j, mode in enumerate(modes):
            # HIP events on the launch stream immediately around the k_render_fwd launch (ops.Render.run)
            if not stub:
                ops.kernel_event_sink = kernel_ms if timed else None
            img = loop(q, t, mode, 1, march, rows=rows)
            if not stub:
                ops.kernel_event_sink = None
            if gather:
                pay_of = lambda: tile_payload(img, mode)
                if stub:
                    pay = pay_of()
                    if pay.shape[1] != per:
                        pay = torch.cat([pay, pay.new_zeros((1, per - pay.shape[1], width, pay.shape[3]))], dim=1)
                    exchange_tile(dist, args, pay, gathered[j] if rank == 0 else None, rank, world, dev)
                    continue
                done = torch.cuda.Event()
                done.record()
                with torch.cuda.stream(comm):
                    comm.wait_event(done)
                    img.record_stream(comm)
                    pay = pay_of()
                    if pay.shape[1] != per:      # ragged last band of a strong-scaling split
                        pay = torch.cat([pay, pay.new_zeros((1, per - pay.shape[1], width, pay.shape[3]))], dim=1)
                    exchange_tile(dist, args, pay, gathered[j] if rank == 0 else None, rank, world, dev)

    def timed_block():
        if comm is not None:
            torch.cuda.current_stream().wait_stream(comm)
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step(True)
        if comm is not None:
            torch.cuda.current_stream().wait_stream(comm)
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    log(f"rank {rank}/{world}: warm-up {args.warmup}, timing {args.repeats} x {args.steps} steps"
        + (f" (config 5: rows {band})" if config5 else ""))
    settle = 0
    with torch.no_grad():
        if config5 and not stub and loop.regen == "auto":
            # RenderLoop's regen="auto" compares its two frame kernels on the first frames of 16-frame cycles;
            # a run of a handful of 65-ms frames would be over before the first comparison
            settle = 36
            for _ in range(settle):
                one_step(False)
                sync()          # every frame waited for, like main.py's window.draw: the choice is made between frames
        for _ in range(args.warmup):
            one_step(False)
        blocks = [timed_block() for _ in range(max(1, args.repeats))]
    elapsed = sorted(blocks)[len(blocks) // 2]                 # median block
    log(f"timed blocks {[round(b, 4) for b in blocks]} s")
    per_launch_ms = sum(a.elapsed_time(b) for a, b in kernel_ms) / max(len(kernel_ms), 1)
    frame_rays = rays_per_frame * (world if not config5 else 1)   # rays of the whole job's frame
    value = frame_rays * len(modes) * args.steps / elapsed / 1e6

    if rank == 0 and stub:
        # launcher / collective rehearsal: nothing was rendered, so there is no rate and no roofline to report
        print(json.dumps({"metric": "STUB: launcher rehearsal, no rendering (bench.py --stub-render)", "stub": True,
                          "value": None, "unit": "Mrays/s", "n_gpus": world, "requested_gpus": args.gpus,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                          "scaling": "strong" if config5 else "weak", "data": "none (host stand-in for the renderer)",
                          "config": {"workload": "stub", "backend": args.backend, "exchange": args.exchange,
                                     "work_per_rank": work_per_rank, "band": list(band),
                                     "gathered_rows": None if gathered is None else per * world}}), flush=True)
    elif rank == 0:
        launch_rays = width * band_rows if world > 1 else rays_per_frame
        evals = launch_rays * (march + 6)
        ach_gbs = launch_rays * BYTES_PER_RAY / (per_launch_ms * 1e-3) / 1e9
        traffic, valu_insts, why = (None, None, "config 5") if config5 else traffic_record(specialised, args)
        if why:
            log(f"roofline.traffic: {why}")
        roof = {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kernel_in_use(loop), "kernel_ms": per_launch_ms,
                **({"kernel_choice": f"regen='auto' after {settle} untimed settling frames, each waited for"} if settle else {}),
                "algorithmic_bytes_per_launch": launch_rays * BYTES_PER_RAY,
                "binding": "fp32-valu issue (SURVEY D8: ~300 flop/B, the fused frame cannot be HBM bound)"}
        if not config5:
            ach_tf = evals * FLOPS_PER_EVAL / (per_launch_ms * 1e-3) / 1e12
            # the VALU headline is what the kernel EXECUTES (SQ_INSTS_VALU from the PMC pass in profiles/):
            # wave-instructions x 64 lanes per second against the 78.6 T lane-instructions/s the 1024 SIMDs issue
            # at the 2-cycle fp32 rate.  The algorithmic figure counts R (S+6) evaluations of 80 flop although the
            # bit-exact early-out and the culling skip ~60 % of them -- it is kept, labelled as such.
            lane_rate = None if not valu_insts else valu_insts * 64 / (per_launch_ms * 1e-3)
            roof["valu"] = {"frac_executed": None if lane_rate is None else lane_rate / (VALU_PEAK_TFLOPS / 2 * 1e12),
                            "executed_wave_instructions": valu_insts,
                            "simd_cycles_per_instruction": (None if not valu_insts else
                                                            per_launch_ms * 1e-3 * 2.4e9 * 1024 / valu_insts),
                            "issue_model": "~4 cycles per instruction is what this mix of compares, selects, min/max, "
                                           "3-operand FMAs and scalar operands can issue (profiles/micro/valu_issue_bench.hip)",
                            "peak_lane_instructions_per_s": VALU_PEAK_TFLOPS / 2 * 1e12,
                            "algorithmic": {"achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": ach_tf / VALU_PEAK_TFLOPS,
                                            "note": "R*(S+6)*80 flop / time: INCLUDES the work the early-out and the "
                                                    "culling skip, i.e. not a utilisation figure"}}
        ms_blocks = [b / args.steps * 1e3 for b in blocks]
        out = {
            "metric": "Mrays/sec at 1920x1080x128 iters; fwd+bwd ms/frame" if not config5
                      else "Mrays/sec at 7680x4320x256 iters, 32-primitive scene (BASELINE configs[4])",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "requested_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if config5 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "blocks": {"repeats": len(blocks), "ms_per_step": ms_blocks, "median": sorted(ms_blocks)[len(ms_blocks) // 2],
                       "min": min(ms_blocks), "max": max(ms_blocks),
                       "note": "value / ms_per_step are the median block; every block is K steps between barrier + synchronise"},
            "config": {"workload": ("BASELINE configs[1]: make_test_scene2 (room shell + sphere/torus/capsule), "
                                    "1920x1080 pinhole ray grid per GPU, 128 march iters, normal + Lambertian "
                                    "frames per step, fp32") if not config5 else
                                   ("BASELINE configs[4]: room + smooth union of 32 affine primitives, 7680x4320 frame, "
                                    "256 march iters, normal shader, fp32, row bands over the ranks"),
                       "frames_per_step": len(modes), "rays_per_frame": rays_per_frame, "camera": [0.0, 0.0, cam_z],
                       "early_out": not args.no_early_out, "wave_tile": "64x1" if args.linear_waves else "8x8", "arithmetic": args.precision,
                       "tile_schedule": "static stride" if args.static_tiles else "64 atomic queues + stealing",
                       "kernels": "per-scene specialised (StaticCfg)" if specialised else "generic LDS interpreter",
                       "parallelism": (f"row-tiles x{world}" if (config5 or banded) else f"camera batch x{world}")
                                      + (f" + {args.exchange} tile exchange to rank 0" if gather else ""),
                       "work_per_rank": work_per_rank,
                       "backend": None if dist is None else ("rccl" if args.backend == "nccl" else args.backend)},
            "ray_sdf_evals_per_s": value * 1e6 * (march + 6),
            "roofline": roof,
        }
        if world == 1 and not config5 and not args.no_pipelined:
            # SURVEY 8(d) names two cameras for config 2, and they are equals here: (0,0,-3), outside the torus,
            # and the reference's own start pose (0,0,1) (main.py:46), inside the torus tube, where more rays
            # never settle.  `value` is the pose of --camera-z (default -3); both are measured the same way.
            # RenderLoop's default regen="auto" times the tile kernel against the ray-regeneration kernels now and then
            # and uses the faster one; the fixed choices are listed beside it.  "per-ray order" deals the rays by the
            # step counts of the previous IDENTICAL frame: right for a viewer that re-renders an unchanged pose, and for
            # this fixed-pose loop -- which rays march long changes with the least camera move, so it is NOT the default.
            poses = []
            variants = {"tile kernel": dict(regen=False), "regeneration, tile-score order": dict(regen=True),
                        "regeneration, per-ray order (unchanged pose only)": dict(regen=True, order_per_ray=True)}
            for z in (-3.0, 1.0):
                tz = torch.tensor([[0.0, 0.0, z]], device=dev)
                auto = make_loop()
                ms = frame_rate(auto, q, tz, rows, frames=32, warm=48)
                entry_ = {"camera": [0.0, 0.0, z], "ms_per_frame": ms, "value": rays_per_frame / ms / 1e3, "unit": "Mrays/s",
                          "kernel": kernel_in_use(auto), "fixed_choices": {},
                          # (frame, pools in use, ms of the probed kernel, ms of the kernel in use) of auto's last decisions
                          "auto_log": [list(x) for st_ in auto._choice_state.values() for x in st_["log"][-6:]],
                          "note": "the reference's default pose (main.py:46)" if z == 1.0 else "outside the torus, every ray hits"}
                del auto
                for name, kw in variants.items():
                    lv = make_loop(**kw)
                    mv = frame_rate(lv, q, tz, rows, frames=32, warm=20)
                    entry_["fixed_choices"][name] = {"ms_per_frame": mv, "value": rays_per_frame / mv / 1e3}
                    del lv
                poses.append(entry_)
            out["poses"] = poses
            out["pipelined"] = pipelined_probe(loop, q, t, rows, dev, rays_per_frame)
            if not args.skip_config3:
                out["config3"] = config3_probe(dev)
            out["reference_shape"] = reference_shape_probe(dev)
            out["first_frames"] = interpreter_record()
        if world == 1 and not args.skip_backward and not config5:      # (single-GPU legs: the other ranks would only wait for rank 0)
            log("backward probe (config 4 shape) ...")
            out["fwd_bwd"] = backward_probe(dev)
            log(f"backward probe: {out['fwd_bwd']}")
        if world == 1 and not args.no_cpu_baseline and not config5:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
