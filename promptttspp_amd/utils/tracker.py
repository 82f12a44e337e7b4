"""Running-mean loss tracking + CSV log (reference: promptttspp/utils/tracker.py)."""
import csv
from collections import OrderedDict


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.sum, self.count = 0.0, 0

    def update(self, value, n=1):
        self.sum += float(value) * n
        self.count += n

    def mean(self):
        return self.sum / max(self.count, 1)


class Tracker:
    def __init__(self, path=None, mode="w"):
        self.path = path
        self.meters = OrderedDict()
        self._header_written = mode == "a"
        if path is not None and mode == "w":
            open(path, "w").close()

    def update(self, **values):
        for k, v in values.items():
            self.meters.setdefault(k, AverageMeter()).update(v)

    def items(self):
        return self.meters.items()

    def write(self, epoch, clear=True):
        if self.path is not None:
            with open(self.path, "a", newline="") as f:
                w = csv.writer(f)
                if not self._header_written:
                    w.writerow(["epoch"] + list(self.meters))
                    self._header_written = True
                w.writerow([epoch] + [f"{m.mean():.6f}" for m in self.meters.values()])
        if clear:
            self.meters = OrderedDict()
