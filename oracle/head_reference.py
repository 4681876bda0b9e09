"""CPU oracle of the head-only training step (TEST INFRASTRUCTURE -- imported by tests/ only, never by fastervit_amd/).

Restates, in plain torch fp32 on the CPU, what the reference computes for the classifier during training:
  * logits = head(features)                         fastervit/models/faster_vit.py:959 (forward_head), head = nn.Linear, :927
  * loss   = LabelSmoothingCrossEntropy(smoothing)  selected at fastervit/train.py:685 (timm 0.9.6, pinned in requirements.txt:1:
             logprobs = log_softmax(x); nll = -logprobs[target]; smooth = -logprobs.mean(-1);
             loss = ((1 - s) * nll + s * smooth).mean());  smoothing = 0 is nn.CrossEntropyLoss (train.py:687)
  * grads  = d loss / d (W, b), averaged over the GLOBAL batch as DistributedDataParallel does (train.py:542-551 all-reduces the
             gradients with op = mean); the loss is averaged across ranks for logging (utils.reduce_tensor, train.py:910)
  * update = SGD with momentum and coupled weight decay (torch.optim.SGD semantics, dampening 0, nesterov False)
Pinned: tests/test_head_train.py checks this closed form against torch.autograd on the same inputs (the autograd graph IS the
reference's arithmetic: F.linear + log_softmax).
"""
import torch


def head_forward_backward_ref(feat, target, weight, bias, global_batch, smoothing):
    """Returns flat [dW | db | loss] for this shard, each already divided by ``global_batch`` (sum over shards = global mean)."""
    feat, weight, bias = feat.double(), weight.double(), bias.double()
    logits = feat @ weight.t() + bias
    logp = torch.log_softmax(logits, dim=-1)
    N = weight.shape[0]
    nll = -logp.gather(1, target.view(-1, 1)).squeeze(1)
    smooth = -logp.mean(dim=-1)
    loss_rows = (1.0 - smoothing) * nll + smoothing * smooth
    q = torch.full_like(logp, smoothing / N)
    q.scatter_add_(1, target.view(-1, 1), torch.full((feat.shape[0], 1), 1.0 - smoothing, dtype=torch.float64))
    dlogits = (logp.exp() - q) / global_batch
    dW = dlogits.t() @ feat
    db = dlogits.sum(0)
    loss = loss_rows.sum() / global_batch
    return torch.cat([dW.reshape(-1), db, loss.view(1)]).float()


def sgd_momentum_ref(param, mom, grad, n, lr, mu, wd):
    g = grad[:n] + wd * param[:n]
    mom[:n] = mu * mom[:n] + g
    param[:n] -= lr * mom[:n]
