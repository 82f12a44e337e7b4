"""Token-budget batching (reference: promptttspp/datasets/utils.py:23-112) -- defines
what ``dataset.max_tokens=30000`` means: greedily grow a batch over length-sorted
indices while (len(batch)+1) * longest_so_far <= max_tokens."""
import random
import sys

from torch.utils.data.sampler import BatchSampler


class ShuffleBatchSampler(BatchSampler):
    def __init__(self, batches, drop_last=False, shuffle=True):
        self.batches, self.drop_last, self.shuffle = batches, drop_last, shuffle

    def __iter__(self):
        if self.shuffle:
            random.shuffle(self.batches)
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def batch_by_size(indices, num_tokens_fn, max_tokens=None, max_sentences=None, required_batch_size_multiple=1):
    max_tokens = sys.maxsize if max_tokens is None else max_tokens
    max_sentences = sys.maxsize if max_sentences is None else max_sentences
    mult = required_batch_size_multiple
    indices = list(indices)
    batches, batch, lens = [], [], []
    longest = 0
    for idx in indices:
        n = num_tokens_fn(idx)
        lens.append(n)
        longest = max(longest, n)
        assert longest <= max_tokens, f"sentence at index {idx} of size {longest} exceeds max_tokens limit of {max_tokens}!"
        full = len(batch) > 0 and (len(batch) == max_sentences or (len(batch) + 1) * longest > max_tokens)
        if full:
            keep = max(mult * (len(batch) // mult), len(batch) % mult)
            batches.append(batch[:keep])
            batch, lens = batch[keep:], lens[keep:]
            longest = max(lens) if lens else 0
        batch.append(idx)
    if batch:
        batches.append(batch)
    return batches
