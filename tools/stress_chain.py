#!/usr/bin/env python3
"""Race hunt for the chained launch (k_net_chain): its inter-workgroup protocol (per-XCD ticket queues, row-tile dependency counters,
same-XCD L2 visibility) replaces launch boundaries, so a protocol error would show as a RARE wrong word, not as an error.  Every
launch here is compared bit for bit with the per-layer launches (MOFA_CHAIN=0) of the same inputs, and the verdict words must say
that every launch was chained, complete and without a timed-out wait.

Arms (shipped fine network 1024 x 10 unless stated):
  * forward, 768 / 512 / 3 row tiles of 256 points (the benchmark's sub-batch, 2/3 of it, a nearly empty chip)
  * the same with a competing stream keeping the CUs unevenly busy
  * two chained launches sharing the chip (two streams, two workspaces: what MOFA_STREAMS=2 does)
  * forward writing the mask tape + the fitting backward's two chained launches (a 1,024-ray fitting step's fine pass), gradients
    compared with the per-layer form's
  * forward keeping the fp32 tape (training forward)
  * the coarse 256 x 8 network's fitting pass: the pipelined persistent kernel writing the mask tape (k_mlp_fused<true>) + its chained backward
  * (round 6) a training step's network pass: tape-keeping forward + the chained training backward (k_net_chain_train), every weight gradient
    compared with the per-layer form's — at the benchmark's sub-batch with a competing stream, on a ragged 6-row-tile batch, on the coarse network

    python tools/stress_chain.py [scale]        # scale 1.0 = >= 5,000 chained launches (about 5 minutes of GPU)
Exit code 1 on any mismatch or verdict.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib, synth  # noqa: E402
from mofanerf_amd.autograd import NetFn, view_bias_torch  # noqa: E402
from mofanerf_amd.hipnet import HipNet  # noqa: E402
from mofanerf_amd.model import NeRF  # noqa: E402

SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
DEV = "cuda"
bad = 0
total_chained = 0


def knob(name, value):
    os.environ[name] = value
    lib.reload_env()


def setup(D, W, R, S, seed=1):
    rng = np.random.default_rng(D + W + R + S)
    net = NeRF(D=D, W=W, input_ch=93, input_ch_views=27, input_ch_textureCodes=256, input_ch_shapeCodes=50, use_viewdirs=True)
    net.load_state_dict(synth.nerf_state(D, W, seed))
    h = HipNet(net.to(DEV))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    o = t(rng.uniform(-2, 2, (R, 3)).astype(np.float32))
    d = t(rng.normal(0, 0.3, (R, 3)).astype(np.float32))
    z = t(np.sort(rng.uniform(8, 26, (R, S)).astype(np.float32), -1))
    vd = torch.nn.functional.normalize(d, dim=-1).contiguous()
    bm, tex, e = synth.codes(3)
    folded = h.fold(e.to(DEV), bm.to(DEV), tex.to(DEV)).clone()
    vb = view_bias_torch(h, vd).detach().contiguous()
    G = t(rng.normal(size=(R, S, 4)).astype(np.float32))
    return h, o, d, z, vd, folded, vb, G


def report(name, launches, mism, t0):
    global bad, total_chained
    bad += mism
    total_chained += launches
    print(f"{name}: {launches} chained launches, {mism} mismatching, {time.perf_counter() - t0:.1f} s", flush=True)


def forward_arm(name, R, S, n, busy=False, two_streams=False, D=10, W=1024):
    h, o, d, z, vd, folded, vb, _ = setup(D, W, R, S)
    knob("MOFA_CHAIN", "0")
    ref = torch.empty(R, S, 4, device=DEV)
    h.forward_rays(o, d, z, S, vd, S, ref, folded)
    torch.cuda.synchronize()
    knob("MOFA_CHAIN", "1")
    before = h.chained_launches()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    outs = [torch.empty(R, S, 4, device=DEV) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    t0, mism, done = time.perf_counter(), 0, 0
    while done < n:
        if busy and done % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    a @ a
        if two_streams:
            main = torch.cuda.current_stream()
            for k, st in enumerate(streams):
                st.wait_stream(main)
                outs[k].fill_(-1.0)
                with torch.cuda.stream(st):
                    h.forward_rays(o, d, z, S, vd, S, outs[k], folded, slot=k)
            for st in streams:
                main.wait_stream(st)
            mism += int(not torch.equal(outs[0], ref)) + int(not torch.equal(outs[1], ref))
            done += 2
        else:
            outs[0].fill_(-1.0)
            h.forward_rays(o, d, z, S, vd, S, outs[0], folded)
            mism += int(not torch.equal(outs[0], ref))
            done += 1
    torch.cuda.synchronize()
    h.check_verdict(block=True)
    assert h.chained_launches() - before == done, "a launch fell back to the per-layer form"
    report(name, done, mism, t0)


def fit_arm(name, R, S, n, fp32_tape=False, D=10, W=1024, per_step=3):
    h, o, d, z, vd, folded, vb, G = setup(D, W, R, S)
    h.force_fp32_tape = fp32_tape

    def step():
        og, dg = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
        fo, vbg = folded.clone().requires_grad_(True), vb.clone().requires_grad_(True)
        raw = NetFn.apply(h, og, dg, z, S, S, fo, vbg, None)
        (raw * G).sum().backward()
        return [raw.detach(), og.grad, dg.grad, fo.grad, vbg.grad]

    knob("MOFA_CHAIN", "0")
    ref = [t.clone() for t in step()]
    torch.cuda.synchronize()
    knob("MOFA_CHAIN", "1")
    before = h.chained_launches()
    t0, mism = time.perf_counter(), 0
    for _ in range(n):
        got = step()
        mism += int(not all(torch.equal(a, b) for a, b in zip(got, ref)))
    torch.cuda.synchronize()
    h.check_verdict(block=True)
    launches = h.chained_launches() - before
    assert launches == per_step * n, f"{launches} chained launches for {n} steps (expected {per_step} each: forward + 2 backward; width 256: the backward only)"
    report(name, launches, mism, t0)


def train_arm(name, R, S, n, D=10, W=1024, busy=False, per_step=3):
    """A training step's network pass (round 6): tape-keeping forward + the chained training backward (k_net_chain_train: backward-data tiles
    AND weight-gradient units behind the queues) — raw, the ray / bias gradients and EVERY weight gradient against the per-layer form."""
    h, o, d, z, vd, folded, vb, G = setup(D, W, R, S)

    def step():
        og, dg = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
        fo, vbg = folded.clone().requires_grad_(True), vb.clone().requires_grad_(True)
        ws = [l.weight.detach().clone().requires_grad_(True) for l in h._linears]
        raw = NetFn.apply(h, og, dg, z, S, S, fo, vbg, None, *ws)
        (raw * G).sum().backward()
        return [raw.detach(), og.grad, dg.grad, fo.grad, vbg.grad] + [w.grad for w in ws]

    knob("MOFA_CHAIN", "0")
    ref = [t.clone() for t in step()]
    torch.cuda.synchronize()
    knob("MOFA_CHAIN", "1")
    knob("MOFA_CHAIN_TRAIN", "1")               # (opt-in: the default training backward is per layer)
    before = h.chained_launches()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    t0, mism = time.perf_counter(), 0
    for i in range(n):
        if busy and i % 2 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    a @ a
        got = step()
        mism += int(not all(torch.equal(x, y) for x, y in zip(got, ref)))
    torch.cuda.synchronize()
    h.check_verdict(block=True)
    launches = h.chained_launches() - before
    knob("MOFA_CHAIN_TRAIN", "0")
    assert launches == per_step * n, f"{launches} chained launches for {n} steps (expected {per_step} each)"
    report(name, launches, mism, t0)


def tape_forward_arm(name, R, S, n, D=10, W=1024):
    """The training forward (fp32 tape), chained: raw and EVERY float of the tape against the per-layer form."""
    h, o, d, z, vd, folded, vb, _ = setup(D, W, R, S)
    Lb, st = lib.load(), lib.stream()
    n_tape = Lb.mofa_net_tape_floats(h.shape, R * S)
    ws = h.workspace(R * S, R, torch.device(DEV, 0))

    def run(tape, raw):
        lib.check(Lb.mofa_net_forward(h.shape, lib.ptr(h.packed()), lib.ptr(folded), None, None, lib.ptr(o), lib.ptr(d), lib.ptr(z), S, None, None, R,
                                      S, lib.ptr(ws), lib.ptr(raw), lib.ptr(tape), None, lib.ptr(vb), h.verdict_ptr(torch.device(DEV, 0)), st), "fwd")

    knob("MOFA_CHAIN", "0")
    ref_t, ref_r = torch.empty(n_tape, device=DEV), torch.empty(R, S, 4, device=DEV)
    run(ref_t, ref_r)
    torch.cuda.synchronize()
    knob("MOFA_CHAIN", "1")
    before = h.chained_launches()
    tape, raw = torch.empty(n_tape, device=DEV), torch.empty(R, S, 4, device=DEV)
    t0, mism = time.perf_counter(), 0
    for _ in range(n):
        tape.fill_(-1.0)
        run(tape, raw)
        mism += int(not (torch.equal(tape, ref_t) and torch.equal(raw, ref_r)))
    torch.cuda.synchronize()
    h.check_verdict(block=True)
    assert h.chained_launches() - before == n
    report(name, n, mism, t0)


def n_of(x):
    return max(2, int(round(x * SCALE)))


print(f"device: {torch.cuda.get_device_name(0)}; XCD census (workgroups of a 2-per-CU launch): {lib.device_init(DEV)}", flush=True)
forward_arm("forward, 768 row tiles (1536 rays x 128: the benchmark's sub-batch)", 1536, 128, n_of(900))
forward_arm("forward, 768 row tiles, competing stream", 1536, 128, n_of(300), busy=True)
forward_arm("forward, 512 row tiles (1024 rays x 128)", 1024, 128, n_of(700))
forward_arm("forward, 512 row tiles, two chained launches sharing the chip (two streams)", 1024, 128, n_of(400), two_streams=True)
forward_arm("forward, 3 row tiles (6 rays x 128)", 6, 128, n_of(1500))
forward_arm("forward, 3 row tiles, competing stream", 6, 128, n_of(500), busy=True)
forward_arm("forward, width 512 x 8, 300 row tiles, competing stream", 1200, 64, n_of(300), busy=True, D=8, W=512)
fit_arm("fitting step fine pass (1024 rays x 128): forward + mask tape, 2 backward launches", 1024, 128, n_of(250))
fit_arm("fitting step, fp32 tape (256 rays x 128)", 256, 128, n_of(60), fp32_tape=True)
fit_arm("coarse network 256 x 8 fitting pass (1024 rays x 64): k_mlp_fused<true> writes the mask tape, 2 chained backward launches", 1024, 64, n_of(200),
        D=8, W=256, per_step=2)
tape_forward_arm("training forward keeping the fp32 tape (512 rays x 128)", 512, 128, n_of(120))
train_arm("training step fine pass (512 rays x 128): tape forward + chained training backward (products + weight gradients)", 512, 128, n_of(60))
train_arm("training step fine pass, 1536 rays x 128 (the benchmark's sub-batch: 768 row tiles), competing stream", 1536, 128, n_of(20), busy=True)
train_arm("training step, 21 rays x 64 of a 512 x 8 network (6 row tiles, ragged last tile)", 21, 64, n_of(300), D=8, W=512)
train_arm("training step coarse pass 256 x 8 (1024 rays x 64): persistent forward, chained training backward", 1024, 64, n_of(100), D=8, W=256, per_step=2)
print(f"TOTAL: {total_chained} chained launches, {bad} mismatching, every verdict clean", flush=True)
sys.exit(1 if bad else 0)
