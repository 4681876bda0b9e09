"""Synthetic stand-ins for timm.data (validate.py:26): a (randn, randint) loader instead of ImageNet."""
import torch


def resolve_data_config(args, model=None, use_test_size=False, verbose=False, **kw):
    cfg = dict(getattr(model, "pretrained_cfg", {}) or {})
    out = {"input_size": tuple(args.get("input_size") or cfg.get("input_size", (3, 224, 224))),
           "interpolation": args.get("interpolation") or cfg.get("interpolation", "bicubic"),
           "mean": args.get("mean") or cfg.get("mean", (0.485, 0.456, 0.406)),
           "std": args.get("std") or cfg.get("std", (0.229, 0.224, 0.225)),
           "crop_pct": args.get("crop_pct") or cfg.get("crop_pct", 0.875),
           "crop_mode": args.get("crop_mode") or cfg.get("crop_mode", "center")}
    return out


class _SyntheticDataset:
    def __init__(self, n=16, num_classes=1000):
        self.n, self.num_classes = n, num_classes

    def __len__(self):
        return self.n

    def filenames(self, basename=False):
        return [f"synthetic_{i}.jpg" for i in range(self.n)]


def create_dataset(name="", root=None, split="validation", **kwargs):
    return _SyntheticDataset()


class _Loader:
    def __init__(self, dataset, input_size, batch_size, device):
        self.dataset, self.input_size, self.batch_size, self.device = dataset, tuple(input_size), batch_size, device

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        g = torch.Generator().manual_seed(0)
        for i in range(len(self)):
            n = min(self.batch_size, len(self.dataset) - i * self.batch_size)
            yield (torch.randn((n,) + self.input_size, generator=g).to(self.device),
                   torch.randint(0, self.dataset.num_classes, (n,), generator=g).to(self.device))


def create_loader(dataset, input_size, batch_size, device=torch.device("cpu"), **kwargs):
    return _Loader(dataset, input_size, batch_size, device)


class RealLabelsImagenet:
    def __init__(self, *a, **k):
        raise NotImplementedError("real labels are not available in the synthetic harness")
