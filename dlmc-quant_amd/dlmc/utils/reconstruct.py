"""FSPTQ block reconstruction with everything resident in HBM.

The reference's loop (`trainer/fsptq_trainer.py:37-100`) hooks a block of the quantised model and of its fp32 twin,
runs the calibration set through both, and copies every hooked activation to the host (`output.cpu()`, :39,42),
concatenates there, copies the result back (:70-71) and then optimises the block on random 64-sample minibatches.
On a 288 GB part the calibration activations of any block fit on the device many times over, so here

  * `collect_block_io` writes the hooked tensors straight into two preallocated device buffers (no host copy, no
    `torch.cat`), and
  * `reconstruct_block` runs the same optimisation (Adam, the reference's per-parameter learning rates, cosine
    schedule, `idx = randperm(n)[:batch]`), with the AdaRound / fake-quant forward and backward as the fused HIP
    kernels of the wrappers.

Both take the modules, not names, so the caller keeps the reference's selection logic (`name in ["conv1",
"linear"]`, block types) unchanged.
"""
import torch

__all__ = ["collect_block_io", "reference_param_groups", "reconstruct_block"]


class _Sink:
    """Preallocated (N, ...) device buffer filled batch by batch from a forward hook."""

    def __init__(self, total):
        self.total, self.buf, self.at = total, None, 0

    def put(self, t):
        t = t.detach()
        if self.buf is None:
            self.buf = torch.empty((self.total,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        n = t.shape[0]
        if self.at + n > self.total:
            raise RuntimeError(f"collect_block_io: {self.at + n} samples hooked, {self.total} announced")
        self.buf[self.at:self.at + n].copy_(t)
        self.at += n


def collect_block_io(model, fp_model, block, fp_block, batches, total):
    """Run `batches` (an iterable of device tensors, `total` samples in all) through both models and return
    `(block_input, block_output)`: the quantised model's input to `block` and the fp32 model's output of
    `fp_block`, as in the reference's `Rhook_quantize` / `Rhook` (fsptq_trainer.py:37-43), kept on the device."""
    xin, yout = _Sink(total), _Sink(total)
    h1 = fp_block.register_forward_hook(lambda m, i, o: yout.put(o))
    h2 = block.register_forward_hook(lambda m, i, o: xin.put(i[0]))
    was = model.training, fp_model.training
    model.eval()
    fp_model.eval()
    try:
        with torch.no_grad():
            for data in batches:
                fp_model(data)
                model(data)
    finally:
        h1.remove()
        h2.remove()
        model.train(was[0])
        fp_model.train(was[1])
    if xin.at != total or yout.at != total:
        raise RuntimeError(f"collect_block_io: hooked {xin.at}/{yout.at} samples, {total} announced")
    return xin.buf, yout.buf


def reference_param_groups(block):
    """The per-parameter learning rates of `FSPTQTrainer.generate_optimizer` (fsptq_trainer.py:136-149)."""
    groups = []
    for name, p in block.named_parameters():
        if not p.requires_grad:
            continue
        if name.endswith("scales"):
            lr = 1e-3
        elif name.endswith("gamma") or name.endswith("beta"):
            lr = 0.1
        else:
            lr = 1e-5
        groups.append({"params": p, "lr": lr})
    return groups


def reconstruct_block(block, block_input, block_output, iters, batch=64, criterion=None, param_groups=None, generator=None,
                      log_every=0, log=print):
    """The optimisation loop of fsptq_trainer.py:77-100 on device-resident activations.  Returns the last loss."""
    if criterion is None:
        criterion = lambda a, b: (a - b).pow(2).mean()  # noqa: E731  trainer/loss/loss.py:22-24 (l2_loss)
    opt = torch.optim.Adam(param_groups if param_groups is not None else reference_param_groups(block))
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=iters, eta_min=0.0)
    n = block_input.shape[0]
    was = block.training
    block.train()
    loss = None
    try:
        for i in range(iters):
            idx = torch.randperm(n, device=block_input.device, generator=generator)[:batch]
            opt.zero_grad(set_to_none=True)
            loss = criterion(block_output[idx], block(block_input[idx]))
            loss.backward()
            opt.step()
            sched.step()
            if log_every and i % log_every == 0:
                log(f"reconstruct_block: iter {i} loss {float(loss):.6f}")
    finally:
        block.train(was)
    return loss
