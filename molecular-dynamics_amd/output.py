"""Asynchronous trajectory output: the step loop captures frames into the engine's pinned ring
and keeps enqueuing; encoder threads wait for each copy to land, compress it and commit the
frames to the files in capture order.

Sized for the production cadence of the pipeline (`nstxout-compressed = 10`,
/root/reference/mdp/production.mdp:12): at ~0.5 ms per step a 140 k-atom frame is due every
~5 ms while its xtc compression takes 3-5 ms on one host core, hence more than one encoder.
The codec runs outside the interpreter lock (ctypes), so the threads do overlap.
"""
from __future__ import annotations

import queue
import threading
import time


class FramePipeline:
    def __init__(self, eng, nslots, xtc, trr, xtc_sel, dt, tinit, nthreads=2):
        self.eng, self.xtc, self.trr, self.sel = eng, xtc, trr, xtc_sel
        self.dt, self.tinit = dt, tinit
        self.q = queue.Queue()
        self.free = queue.Queue()
        for s in range(nslots):
            self.free.put(s)
        self.error = None
        # where the encoder threads' time goes (seconds, summed over frames): waiting for the copy to land, compressing,
        # waiting for the turn to commit plus the write itself
        self.stats = {"frames": 0, "wait_s": 0.0, "encode_s": 0.0, "commit_s": 0.0}
        self._seq = 0
        self._next_commit = 0
        self._cv = threading.Condition()
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(max(1, nthreads))]
        for t in self.threads:
            t.start()

    def acquire_slot(self):
        """Blocks while every slot still waits for its encoder (back-pressure on the step loop)."""
        return self.free.get()

    def capture(self, want_xtc, want_trr_x, want_trr_v):
        """Take a free slot, enqueue the capture of the current state into it and hand it to the encoders. A frame that goes
        to the compressed trajectory only, whole system, is rounded to the xtc integers on the GPU (the encoder thread is
        left with the bit packing); every other frame arrives as floats."""
        slot = self.acquire_slot()
        quant = bool(want_xtc and self.xtc is not None and self.sel is None and not want_trr_x and self.xtc.natoms > 9)
        if quant:
            self.eng.capture_frame_xtc(slot, self.xtc.precision, want_v=want_trr_v)
        else:
            self.eng.capture_frame(slot, want_v=want_trr_v)
        self.q.put((self._seq, slot, want_xtc, want_trr_x, want_trr_v, quant))
        self._seq += 1

    def submit(self, slot, want_xtc, want_trr_x, want_trr_v):
        """A slot the caller has captured into with Engine.capture_frame (floats)."""
        self.q.put((self._seq, slot, want_xtc, want_trr_x, want_trr_v, False))
        self._seq += 1

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            seq, slot, wx, tx, tv, quant = item
            xb = None
            args = None
            t_a = t_b = t_c = time.perf_counter()
            try:
                if quant:
                    step, ix, v, box, stats = self.eng.frame_wait_xtc(slot)
                    x = None
                else:
                    step, x, v, box = self.eng.frame_wait(slot)
                t_b = time.perf_counter()
                t = self.tinit + step * self.dt
                if quant:
                    xb = self.xtc.encode_quantized(step, t, box, ix, stats)
                elif wx and self.xtc is not None:
                    xb = self.xtc.encode(step, t, box, x if self.sel is None else x[self.sel])
                t_c = time.perf_counter()
                if (tx or tv) and self.trr is not None:
                    args = (step, t, box.copy(), x.copy() if tx else None, v.copy() if (tv and v is not None) else None)
            except Exception as e:      # keep draining so that the step loop never blocks on a slot
                self.error = self.error or e
            finally:
                self.free.put(slot)
            with self._cv:
                while self._next_commit != seq:
                    self._cv.wait()
                try:
                    if xb is not None:
                        self.xtc.write_bytes(xb)
                    if args is not None:
                        self.trr.write(*args)
                except Exception as e:
                    self.error = self.error or e
                self._next_commit += 1
                st = self.stats
                st["last"] = (t_a, t_b, t_c, time.perf_counter())     # taken from the queue, copy landed, packed, committed
                st["frames"] += 1; st["wait_s"] += t_b - t_a; st["encode_s"] += t_c - t_b; st["commit_s"] += time.perf_counter() - t_c
                self._cv.notify_all()

    def wait_idle(self):
        """Blocks until every frame submitted so far has been encoded and committed; the threads stay up."""
        with self._cv:
            while self._next_commit != self._seq:
                self._cv.wait()

    def pending(self):
        """Frames submitted and not yet committed."""
        return self._seq - self._next_commit

    def finish(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
