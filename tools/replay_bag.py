#!/usr/bin/env python
"""Replay a ROS bag (format 2.0) through the GPU path:  python tools/replay_bag.py FR_IOSB_Short.bag [--lidar /livox/lidar] [--imu /imu]
Prints one line per scan: stamp, position, quaternion (w x y z), feature counts.  Needs an MI355X."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lili_om_amd as L
from lili_om_amd import replay

ap = argparse.ArgumentParser()
ap.add_argument("bag")
ap.add_argument("--lidar", default="/livox/lidar")
ap.add_argument("--imu", default="/livox/imu")
ap.add_argument("--max-scans", type=int, default=0)
a = ap.parse_args()
ctx = L.Context(0)
replay.replay(a.bag, ctx, a.lidar, a.imu, max_scans=a.max_scans or None,
              on_frame=lambda r: print(f"{r['stamp']:.6f} " + " ".join(f"{v:.6f}" for v in list(r['t']) + list(r['q'])) + f" surf {r['n_surf']} edge {r['n_edge']} queries {r['n_query']}", flush=True))
ctx.close()
