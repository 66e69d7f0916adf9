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
        try:
            m = self.model
            with torch.no_grad():                          # evaluate.cpp:221
                while self.cur_prog < len(out) and not self.aborted:
                    self.status = "inferencing"
                    for i, io in enumerate(out[self.cur_prog]):
                        io = np.ascontiguousarray(io, dtype=np.float32)
                        if io.ndim != 3 or io.shape[0] % m.in_count:
                            raise E.UNetError("model_io buffer must be (in_count*D, H, W), got %s" % (io.shape,))
                        d = io.shape[0] // m.in_count
                        x = torch.from_numpy(io).view(1, m.in_count, d, io.shape[1], io.shape[2]).to(self.device)
                        result = m.forward(x)[0]                                         # evaluate.cpp:226-227
                        out[self.cur_prog][i] = result.to("cpu").contiguous().numpy().reshape(m.out_count * d, io.shape[1], io.shape[2])
                    self.cur_prog += 1
        except Exception as e:                                                           # evaluate.cpp:234-242
            self.error_msg = "error during evaluation:" + str(e)
            self.aborted = True
        self.running = False
        return out
