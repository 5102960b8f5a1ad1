"""LQR closed loop (closed_loop_prediction as one persistent kernel): one agent per lane against one agent per DPP quad, through
crx_x_lqr_closed_loop_lanes_dev, on side_bench's workload (the reference course, episodes of up to 400 ticks).  One JSON line per
(dim, n); final states and tick counts of the two layouts are compared bit for bit."""
import json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpprobotics_amd as crx
from cpprobotics_amd.experimental import closed_loop_prediction_lanes
from common import lqr_course, tracking_agents
course, goal = lqr_course()
dc = crx.Course.from_numpy(course)
max_ticks = 400
for dim in (5, 4):
    for n in (4096, 16384, 32768, 65536, 262144):
        st = tracking_agents(n, tuple(c[:200] for c in course), 5, spread=0.4)
        std = torch.from_numpy(st).cuda()
        out = {}
        for lanes in (1, 4):
            sd = std.clone()
            ticks, _ = closed_loop_prediction_lanes(sd, dc, goal, lanes, dim=dim, max_ticks=max_ticks)
            torch.cuda.synchronize()
            ms = []
            for _ in range(3):
                s2 = std.clone()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); closed_loop_prediction_lanes(s2, dc, goal, lanes, dim=dim, max_ticks=max_ticks); b.record()
                torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
            out[lanes] = (min(ms), sd.cpu().numpy(), ticks.cpu().numpy())
        tk = out[1][2].astype(np.int64)
        same = bool(np.array_equal(out[1][1].view(np.uint32), out[4][1].view(np.uint32)) and np.array_equal(out[1][2], out[4][2]))
        print(json.dumps({"dim": dim, "agents": n, "agent_ticks": int(tk.sum()), "ms_one_lane": out[1][0], "ms_quad": out[4][0],
                          "agent_ticks_per_s_one_lane": tk.sum() / out[1][0] * 1e3, "agent_ticks_per_s_quad": tk.sum() / out[4][0] * 1e3,
                          "quad_speedup": out[1][0] / out[4][0], "bit_identical": same}), flush=True)
