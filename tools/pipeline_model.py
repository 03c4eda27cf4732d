#!/usr/bin/env python3
"""Discrete-event model of abea_align_batch_host on BASELINE configs[2] (CPU only): what launch order, chunk
size and slot count do to a step when the host loops and the GPU overlap.

  host   one caller thread: per chunk  [retire chunk c - slots: wait for its last read, un-flatten] -> plan -> flatten -> enqueue
         (rates measured on the MI355X host: flatten 10.6 Gevents/s, pair expansion 45 Gevents/s, 100 ns of planning per read)
  GPU    4096 wave slots, one read per slot, reads dispatched in enqueue order; every resident wave advances at
         1 / lat(n) bands per ns, lat(n) = 225 ns (lone wave) .. LATF (4096 resident) — LATF calibrated so that the whole batch at
         full occupancy takes the measured 340 ms (align-pre + alignment kernels)

It reproduced the measured 367-370 ms of the longest-first order with 16 hardware queues and predicted 349-352 ms for the
ascending ramp that ships (measured: 357-363 ms).  python tools/pipeline_model.py
"""
import heapq
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5c_amd import synth

SLOTS, LAT1, GPU_MS = 4096, 225.0, 340.0


class Gpu:
    def __init__(self, latf):
        self.t = 0.0; self.V = 0.0; self.res = []; self.queue = []; self.qpos = 0; self.done = {}; self.left = {}; self.latf = latf

    def lat(self, n):
        return LAT1 + (self.latf - LAT1) * min(n, SLOTS) / SLOTS

    def enqueue(self, t, chunk, bands):
        self.advance(t)
        self.left[chunk] = len(bands)
        self.queue += [(chunk, b) for b in bands]
        self.fill()

    def fill(self):
        while len(self.res) < SLOTS and self.qpos < len(self.queue):
            c, b = self.queue[self.qpos]; self.qpos += 1
            heapq.heappush(self.res, (self.V + b, c))

    def advance(self, t_target):
        while self.t < t_target:
            if not self.res:
                self.t = t_target
                break
            lat = self.lat(len(self.res))
            dt = (self.res[0][0] - self.V) * lat * 1e-6
            if self.t + dt <= t_target:
                self.t += dt; self.V = self.res[0][0]
                while self.res and self.res[0][0] <= self.V + 1e-9:
                    _, c = heapq.heappop(self.res)
                    self.left[c] -= 1
                    if self.left[c] == 0:
                        self.done[c] = self.t
                self.fill()
            else:
                self.V += (t_target - self.t) * 1e6 / lat; self.t = t_target

    def wait(self, c, t_now):
        self.advance(t_now)
        while c not in self.done:
            lat = self.lat(len(self.res))
            self.advance(self.t + (self.res[0][0] - self.V) * lat * 1e-6 + 1e-9)
        return max(t_now, self.done[c])


def carve(E, order, chunk_events=48 << 20, rmin=2048, rmax=16384, ramp=(4, 2)):
    out, pos = [], 0
    while pos < len(order):
        r = ramp[len(out)] if len(out) < len(ramp) else 1
        ev, end = 0, pos
        while end < len(order):
            ev += E[order[end]]; end += 1
            if (end - pos >= max(1, rmin // r) and ev >= chunk_events / r) or end - pos >= rmax:
                break
        out.append(order[pos:end]); pos = end
    return out


def step_ms(E, B, order, F=10.6e6, U=45e6, plan_ns=100, setup=2.0, slots=8, **kw):
    chunks = carve(E, order, **kw)
    g = Gpu(GPU_MS * 1e6 * SLOTS / B.sum())
    t, inflight = setup, []
    for ci, ch in enumerate(chunks):
        if len(inflight) == slots:
            c0 = inflight.pop(0)
            t = g.wait(c0, t) + E[chunks[c0]].sum() / U
        t += len(ch) * plan_ns * 1e-6 + E[ch].sum() / F
        g.enqueue(t + 0.5, ci, B[ch])
        inflight.append(ci)
    for c0 in inflight:
        t = g.wait(c0, t) + E[chunks[c0]].sum() / U
    return t


def ramp_order(E, B, floor_bands=18000, every=2):
    o = np.argsort(-B, kind="stable")
    p = int(np.searchsorted(-B[o], -floor_bands, side="right"))
    take = np.zeros(p, bool); take[1::every] = True
    rest = np.ones(len(o), bool); rest[:p] = ~take
    return np.concatenate([o[:p][take][::-1], o[rest]])


if __name__ == "__main__":
    cfg = synth.CONFIGS["r9_100k_mixed"]
    L = synth.batch_lengths(cfg["n_reads"], cfg["seed"], cfg["law"]).astype(np.int64)
    E = (2.04 * L).astype(np.int64); B = E + (L - 5) + 2
    lpt = np.argsort(-B, kind="stable")
    print(f"{len(L)} reads, {E.sum() / 1e9:.2f} G events, {B.sum() / 1e9:.2f} G bands; GPU alone {GPU_MS:.0f} ms")
    print(f"longest-first                         {step_ms(E, B, lpt):6.1f} ms")
    print(f"ascending ramp (ships)                {step_ms(E, B, ramp_order(E, B)):6.1f} ms")
    print(f"ramp of every 3rd read                {step_ms(E, B, ramp_order(E, B, every=3)):6.1f} ms")
    print(f"ramp, 16 slots                        {step_ms(E, B, ramp_order(E, B), slots=16):6.1f} ms")
    print(f"ramp, 24 M-event chunks               {step_ms(E, B, ramp_order(E, B), chunk_events=24 << 20, rmin=1024):6.1f} ms")
    print(f"longest-first, host as in round 4     {step_ms(E, B, lpt, F=9.5e6, U=42e6, plan_ns=240, setup=10):6.1f} ms   (ideal GPU sharing: the 428.7 ms "
          "measured then were the hardware-queue serialisation, which this model does not have)")
