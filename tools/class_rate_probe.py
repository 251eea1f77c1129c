#!/usr/bin/env python3
"""Scratch probe: steady-state cost of the 16-bit wavefront kernels per read length (one class per run, 1 M reads vs a 400 bp window):
ms per million reads, and ns per (read x column x segment) of the forward pass -- flat across classes when occupancy does not matter."""
import sys, time
sys.path.insert(0, ".")
import indelpost_amd as ip
from indelpost_amd import synth

routing = int(sys.argv[1]) if len(sys.argv) > 1 else 0          # library routing switches (indelpost_amd.batch.ROUTE_*)
lens = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100, 125, 150, 200, 250]
for rl in lens:
    jobs = synth.config2_jobs(1_000_000, 3, 1, rl, 400)
    g = ip.MultiStreamAligner(0, 3, 2, streams=4)
    if routing:
        g.set_routing(routing)
    g.upload(jobs)
    for _ in range(2):
        g.run()
    g.sync()
    g.set_profiling(2)
    t0 = time.perf_counter()
    K = 4
    for _ in range(K):
        g.run()
    g.sync()
    dt = (time.perf_counter() - t0) / K
    kt = g.kernel_times()
    S = (rl + 7) // 8
    fw = sum(v[0] for k, v in kt.items() if k.startswith("dp_word_first")) / K / 4      # per stream-launch sum -> ms of stream time per step / 4 streams
    rv = sum(v[0] for k, v in kt.items() if k.startswith("dp_word_rev")) / K / 4
    print("read %3d bp (segLen %2d): %.2f ms/step = %.1f M aln/s | fwd %.2f ms rev %.2f ms (stream time / 4) | fwd ns per read*column*segment %.4f" % (
        rl, S, dt * 1e3, 1.0 / dt, fw, rv, fw * 1e6 / (1e6 * 407 * S)), flush=True)
    g.close()
