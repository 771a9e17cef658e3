"""Teacher-forced NLL on the generator kernel (metrics/nll.py:49-63)."""
import torch

from . import ops


def cal_nll(model, data_loader, criterion=None, gpu=True, precision=None):
    """NLL.cal_nll (metrics/nll.py:49-63): mean over batches of the mean token NLL, rounded to 4
    places.  ``data_loader`` yields dicts with 'input' / 'target' [B,L] like GenDataIter.loader
    (utils/data_loader.py:38-42); ``criterion`` is accepted for signature parity (the reference
    passes nn.NLLLoss(), i.e. a mean) and ignored.  The vocabulary projection, log-softmax and the
    target pick are fused in the kernel: no [B*L, V] matrix is materialised.  ``precision``: "fp32" or
    "split_f16" (None = the library default, _lib.default_gen_precision); see tg_gen_nll in the header."""
    dev = model.lstm2out.weight.device
    losses = []
    with torch.no_grad():
        for data in data_loader:
            inp, target = data["input"].to(dev), data["target"].to(dev)
            logp = ops.gen_nll(model, inp, target, precision)
            losses.append(-logp.mean())
        if not losses:
            raise ZeroDivisionError("empty data loader")  # the reference divides by len(data_loader)
        total = torch.stack(losses).double().sum().item()
    return round(total / len(losses), 4)


def nll_of_samples(model, samples, batch_size, start_letter=1, precision=None):
    """cal_nll over an in-memory [N,L] sample tensor without a DataLoader: one launch for all full
    batches (drop_last like utils/data_loader.py:42), GenDataIter.prepare's shift
    (utils/data_loader.py:70-80) done on the device.  Returns (rounded nll, per-batch losses)."""
    dev = model.lstm2out.weight.device
    s = samples.to(dev).long()
    nb = s.shape[0] // batch_size
    if nb == 0:
        raise ZeroDivisionError("fewer samples than one batch")
    s = s[:nb * batch_size]
    inp = torch.empty_like(s)
    inp[:, 0] = start_letter
    inp[:, 1:] = s[:, :-1]
    with torch.no_grad():
        logp = ops.gen_nll(model, inp, s, precision)
        per_batch = -logp.view(nb, -1).mean(1)
        total = per_batch.double().sum().item()
    return round(total / nb, 4), per_batch
