"""Inference loop of the reference over the engine: `evaluate_unet::start()` -> `prepare_for_inference`, then
`evaluate_unet::evaluate()` (evaluate.cpp:386-399, 211-246) for volumes that are already pre-processed host buffers
(reading, pre-/post-processing and file output are TIPL code and out of scope).

One `model_io` buffer is a float32 host array of shape (in_count*D, H, W): the input channels stacked along z
(evaluate.cpp:226-227).  After the forward it holds (out_count*D, H, W): the full-resolution logits [0] of the network, copied
back to the host (evaluate.cpp:228-229).  Errors do not propagate: like the reference's thread they set `error_msg` and `aborted`
(evaluate.cpp:234-242)."""
import numpy as np
import torch

from . import engine as E


class EvaluateUNet:
    def __init__(self, model, device=None):
        self.model = model
        self.device = torch.device(device) if device is not None else model.device()
        self.error_msg = ""
        self.aborted = False
        self.running = False
        self.cur_prog = 0
        self.status = ""

    def start(self, model_io):
        """model_io: list (one entry per input file) of lists of host buffers (evaluate.hpp:18: eval[i].model_io).
        Returns the same structure with every buffer replaced by its (out_count*D, H, W) result."""
        self.status = "initiating"
        self.model.prepare_for_inference(self.device)     # evaluate.cpp:391
        self.aborted, self.running, self.error_msg, self.cur_prog = False, True, "", 0
        out = [list(ios) for ios in model_io]
        pending = None      # (file index, buffer index, pinned host tensor, shape, event): the previous result, still in flight
        copy_stream = torch.cuda.Stream(self.device)

        def land(p):
            fi, bi, host, shape, ev = p
            ev.synchronize()
            out[fi][bi] = host.numpy().reshape(shape)   # a view of the pinned buffer the copy landed in (owned by the array)

        try:
            m = self.model
            packed_sizes = set()                           # volume sizes whose filter packs this run has already made (weights are frozen)
            with torch.no_grad():                          # evaluate.cpp:221
                while self.cur_prog < len(out) and not self.aborted:
                    self.status = "inferencing"
                    for i, io in enumerate(out[self.cur_prog]):
                        io = np.ascontiguousarray(io, dtype=np.float32)
                        if io.ndim != 3 or io.shape[0] % m.in_count:
                            raise E.UNetError("model_io buffer must be (in_count*D, H, W), got %s" % (io.shape,))
                        d = io.shape[0] // m.in_count
                        x = torch.from_numpy(io).view(1, m.in_count, d, io.shape[1], io.shape[2]).to(self.device)
                        size = tuple(x.shape[2:])
                        result = m.forward(x, packs_current=size in packed_sizes)[0]     # evaluate.cpp:226-227
                        packed_sizes.add(size)
                        # evaluate.cpp:228-229 copies the logits to the host before the next forward starts; here the copy runs on its
                        # own stream into pinned memory under the next buffer's upload + forward (same bytes, same order of results)
                        done = torch.cuda.Event()
                        done.record(torch.cuda.current_stream(self.device))
                        # (pinning costs ~2.6 ms per 50 MB result whether it is done per volume or in one arena for the whole run:
                        # measured 4.5 vs 7.1 ms per volume, profiles/bench_evaluate.py)
                        host = torch.empty(result.shape, dtype=torch.float32, pin_memory=True)
                        with torch.cuda.stream(copy_stream):
                            copy_stream.wait_event(done)
                            host.copy_(result, non_blocking=True)
                            result.record_stream(copy_stream)
                            ev = torch.cuda.Event()
                            ev.record(copy_stream)
                        if pending is not None:
                            land(pending)
                        pending = (self.cur_prog, i, host, (m.out_count * d, io.shape[1], io.shape[2]), ev)
                    self.cur_prog += 1
                if pending is not None:
                    land(pending)
                    pending = None
        except Exception as e:                                                           # evaluate.cpp:234-242
            self.error_msg = "error during evaluation:" + str(e)
            self.aborted = True
        self.running = False
        return out
