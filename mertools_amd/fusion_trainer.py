"""hipGraph-replayed fusion training step.

The reference's inner loop (main-release.py:31-66) issues ~60 tiny kernels per minibatch and syncs the host 2-5 times;
at 445k parameters it is pure launch latency.  Here one optimiser step — forward, CE + MSE losses, backward, value
clipping and Adam, all on the HIP kernels of libmer_hip.so — is captured ONCE into a HIP graph (torch.cuda.CUDAGraph
records whatever is launched on the capture stream, which is where mertools_amd.ops launches) and replayed per
minibatch: the host cost of a step is one graph launch plus the input copies, and nothing syncs until the caller reads
a result.  Parameters stay ordinary nn.Parameters (views of one flat buffer), so state_dict()/checkpoints interoperate
with the reference's `Attention` module.
"""
import torch

from . import _lib
from .ops import _p, stream
from .toolkit.utils.loss import CELoss, MSELoss


class GraphAdam:
    """torch.optim.Adam semantics over ONE flat parameter buffer, step counter on the device (graph-replayable)."""

    def __init__(self, flat_param, flat_grad, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=-1.0):
        self.p, self.g = flat_param, flat_grad
        self.m, self.v = torch.zeros_like(flat_param), torch.zeros_like(flat_param)
        self.step_dev = torch.zeros((), dtype=torch.int32, device=flat_param.device)
        self.hp = (lr, betas[0], betas[1], eps, weight_decay, float(clip_value))

    def step(self):
        lr, b1, b2, eps, wd, clip = self.hp
        lib = _lib.lib()
        _lib.check(lib.mer_adam_step_dev(_p(self.p), _p(self.g), _p(self.m), _p(self.v), self.p.numel(), lr, b1, b2, eps, wd,
                                         _p(self.step_dev), clip, stream()), "mer_adam_step_dev")
        _lib.check(lib.mer_inc_i32(_p(self.step_dev), stream()), "mer_inc_i32")


class FusionGraphTrainer:
    """Wraps a fusion model (toolkit.models.get_models(args) or Attention/LF_DNN) for graph-replayed training steps.

        tr = FusionGraphTrainer(model, lr=1e-3, weight_decay=1e-5, grad_clip=-1.0)
        loss, emos_out, vals_out = tr.train_step(batch, emos, vals)      # device tensors, no host sync
    """

    def __init__(self, model, lr, weight_decay=0.0, grad_clip=-1.0, use_graph=True):
        self.model = model
        params = [p for p in model.parameters() if p.requires_grad]
        dev = params[0].device
        if not params[0].is_cuda:
            raise _lib.MerError("FusionGraphTrainer needs the model on the GPU (there is no CPU path)")
        n = sum(p.numel() for p in params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:  # re-point every parameter (and its .grad) at a slice of the flat buffers
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
            p.grad = self.flat_grad[off:off + k].view_as(p.data)
            off += k
        self.opt = GraphAdam(self.flat, self.flat_grad, lr, weight_decay=weight_decay, clip_value=grad_clip)
        self.cls_loss, self.reg_loss = CELoss(), MSELoss()
        self.use_graph = use_graph
        self._graphs = {}

    def _step_body(self, batch, emos, vals):
        self.flat_grad.zero_()
        features, emos_out, vals_out, interloss = self.model(batch)
        loss = interloss + self.cls_loss(emos_out, emos) + self.reg_loss(vals_out, vals)
        loss.backward()   # autograd accumulates into the flat_grad views (p.grad is kept, never re-allocated)
        self.opt.step()
        return loss, emos_out, vals_out

    @torch.no_grad()
    def _eval_body(self, batch, emos, vals):
        features, emos_out, vals_out, interloss = self.model(batch)
        loss = interloss + self.cls_loss(emos_out, emos) + self.reg_loss(vals_out, vals)
        return loss, emos_out, vals_out

    def _capture(self, shapes, train):
        """shapes: {'audios': (B, ...), 'texts': ..., 'videos': ...} — utterance-level [B, D] or frame-level [B, T, D] inputs."""
        dev = self.flat.device
        B = shapes["audios"][0]
        st = {k: torch.zeros(shapes[k], device=dev) for k in ("audios", "texts", "videos")}
        st["emos"] = torch.zeros(B, dtype=torch.int64, device=dev)
        st["vals"] = torch.zeros(B, device=dev)
        body = self._step_body if train else self._eval_body
        self.model.train() if train else self.model.eval()
        # warm up on a side stream (torch's capture protocol), restoring parameters/optimiser state afterwards
        snap = (self.flat.clone(), self.opt.m.clone(), self.opt.v.clone(), self.opt.step_dev.clone())
        rng = torch.cuda.get_rng_state(dev)

        def restore():
            self.flat.copy_(snap[0]); self.opt.m.copy_(snap[1]); self.opt.v.copy_(snap[2]); self.opt.step_dev.copy_(snap[3])

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                body({k: st[k] for k in ("audios", "texts", "videos")}, st["emos"], st["vals"])
        torch.cuda.current_stream().wait_stream(s)
        restore()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = body({k: st[k] for k in ("audios", "texts", "videos")}, st["emos"], st["vals"])
        restore()
        torch.cuda.set_rng_state(rng, dev)   # the warm-up's dropout draws must not shift the caller's random stream
        return g, st, out

    def _replay(self, batch, emos, vals, train):
        key = (train,) + tuple(tuple(batch[k].shape) for k in ("audios", "texts", "videos"))
        if key not in self._graphs:
            self._graphs[key] = self._capture({k: tuple(batch[k].shape) for k in ("audios", "texts", "videos")}, train)
        g, st, out = self._graphs[key]
        for k in ("audios", "texts", "videos"):
            st[k].copy_(batch[k], non_blocking=True)     # host (pinned) or device source
        st["emos"].copy_(emos, non_blocking=True)
        st["vals"].copy_(vals, non_blocking=True)
        self.model.train() if train else self.model.eval()
        g.replay()
        return out

    def train_step(self, batch, emos, vals):
        """One optimiser step.  Returns (loss, emos_out, vals_out): with use_graph these are the graph's STATIC output buffers —
        copy what you keep before the next step."""
        self.model.train()
        if not self.use_graph:
            dev = self.flat.device
            return self._step_body({k: v.to(dev, non_blocking=True) for k, v in batch.items()}, emos.to(dev, non_blocking=True),
                                   vals.to(dev, non_blocking=True))
        return self._replay(batch, emos, vals, True)

    def eval_step(self, batch, emos=None, vals=None):
        """Forward (+ losses when labels are given) without touching parameters; graph-replayed like train_step."""
        self.model.eval()
        if emos is None:
            with torch.no_grad():
                return self.model(batch)
        if not self.use_graph:
            dev = self.flat.device
            return self._eval_body({k: v.to(dev, non_blocking=True) for k, v in batch.items()}, emos.to(dev, non_blocking=True),
                                   vals.to(dev, non_blocking=True))
        return self._replay(batch, emos, vals, False)
