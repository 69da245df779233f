"""Pool statistics of k_march_regen (library built with -DRM_REGEN_STATS): 4-step groups walked, lanes busy in them,
refills, and the same after the queues ran dry.  usage: RM_HIPCC_EXTRA=-DRM_REGEN_STATS python profiles/regen_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd import ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
dev = torch.device("cuda:0")
h, w = 1080, 1920
q = torch.tensor([[1.0, 0, 0, 0]], device=dev)
for z in (1.0, -3.0):
    for order in (0, 16):
        loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                          sensor_height=bench.PX * h, normals_eps=bench.EPS, regen=True, adaptive_order=order).to(dev)
        t = torch.tensor([[0.0, 0.0, z]], device=dev)
        ops.kernel_event_sink = []
        with torch.no_grad():
            for _ in range(3):
                loop(q, t, 4, 1, 128)
        torch.cuda.synchronize()
        wk = ops.fwd_last_work.cpu().tolist()
        g, l, r, tg, tl, mg, mtg = wk[8:15]
        ev = ops.kernel_event_sink[-1]
        print(f"z={z:+g} order={order}: frame {1e3 * ev[0].elapsed_time(ev[1]):.0f} us; groups {g}, busy lanes/group {l / max(g, 1):.1f}, "
              f"refills {r}; after the queues ran dry: groups {tg} ({100.0 * tg / max(g, 1):.0f} %), busy lanes/group {tl / max(tg, 1):.1f}; "
              f"longest wave {mg} groups, of which dry {mtg}; lane-steps {4 * l / 1e6:.1f} M")
        ops.kernel_event_sink = None
        for key, st in loop._order_state.items():
            if st.get("T"):
                c, o = st["cost"].cpu(), st["order"].cpu()
                if st["score"] is not None:
                    sc = st["score"].cpu()
                    tmax = c.view(-1, 64).max(dim=1).values
                    nlong = (c.view(-1, 64) >= 96).sum(dim=1)
                    dec = [int(i * (len(o) - 1) / 10) for i in range(11)]
                    print("   tile order: score along the order (deciles)", [int(sc[o[i]]) for i in dec], "longest ray of the tile", [int(tmax[o[i]]) for i in dec],
                          "long rays in the tile", [int(nlong[o[i]]) for i in dec], "tiles with a long ray", int((nlong > 0).sum()), "of", len(o))
                else:
                    print("   ray order: cost along the order (deciles)", [int(c[o[int(i * (len(o) - 1) / 10)]]) for i in range(11)])
