"""Host-side cost of one eager RenderLoop.forward call (tiny frame so the GPU is never the limiter)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX = 3.45e-6
dev = torch.device("cuda:0")
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=64, px_height=64, focal_length=PX*64, sensor_width=PX*64, sensor_height=PX*64, normals_eps=5e-2).to(dev)
q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, -3.0]], device=dev)
with torch.no_grad():
    for _ in range(50): loop(q, t, 4, 1, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): loop(q, t, 4, 1, 8)
    torch.cuda.synchronize()
    print(f"eager: {(time.perf_counter()-t0)/2000*1e6:.1f} us per call")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): loop(q, t, 4, 1, 8)
    pr.disable()
    frame = loop.capture(4, 1, 8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2000): frame(q, t)
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter()-t0)/2000*1e6:.1f} us per call")
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(18)
