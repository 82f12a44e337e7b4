"""Host-side helpers with the reference's names (promptttspp/utils/model.py)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def remove_weight_norm_(m):
    """``module.apply(remove_weight_norm_)`` helper (utils/model.py:23-27)."""
    try:
        nn.utils.remove_weight_norm(m)
    except ValueError:  # module without weight norm
        return


def sequence_mask(length, max_length=None):
    """(B,) lengths -> (B, max) bool prefix mask (utils/model.py:30-34)."""
    if max_length is None:
        max_length = length.max()
    steps = torch.arange(int(max_length), dtype=length.dtype, device=length.device)
    return steps.unsqueeze(0) < length.unsqueeze(1)


def frame_to_phone_index(duration, n_frames):
    """Integer alignment: index of the phone that owns each frame, -1 past the end.
    duration: (B, Tp) integer frames per phone.  This is what the HIP length
    regulator computes on the fly (ptpp_length_regulate_fwd)."""
    cum = torch.cumsum(duration.long(), dim=1)
    f = torch.arange(n_frames, device=duration.device)
    idx = torch.searchsorted(cum, f.unsqueeze(0).expand(cum.shape[0], -1).contiguous(), right=True)
    return torch.where(idx < duration.shape[1], idx, torch.full_like(idx, -1))


def generate_path(duration, mask):
    """Dense 0/1 monotonic alignment (B, Tp, Tf) -- kept for API compatibility
    (utils/model.py:37-47); the model itself uses the gather form above."""
    cum = torch.cumsum(duration, dim=1)
    lo = F.pad(cum, (1, 0))[:, :-1]
    f = torch.arange(mask.shape[2], dtype=cum.dtype, device=duration.device)[None, None, :]
    inside = (f < cum.unsqueeze(-1)) & ~(f < lo.unsqueeze(-1))
    return inside.to(mask.dtype) * mask


def to_log_scale(x):
    """log of the non-zero entries, IN PLACE like the reference (utils/model.py:62-64)."""
    nz = x != 0
    x[nz] = torch.log(x[nz])
    return x


def make_pad_mask(lengths, xs=None, length_dim=-1, maxlen=None):
    if length_dim == 0:
        raise ValueError("length_dim cannot be 0: {}".format(length_dim))
    if not isinstance(lengths, torch.Tensor):
        lengths = torch.as_tensor(lengths)
    bs = lengths.shape[0]
    if maxlen is None:
        maxlen = int(lengths.max()) if xs is None else xs.size(length_dim)
    steps = torch.arange(0, maxlen, dtype=torch.int64, device=lengths.device)
    mask = steps.unsqueeze(0).expand(bs, maxlen) >= lengths.reshape(bs, 1)
    if xs is not None:
        assert xs.size(0) == bs, (xs.size(0), bs)
        if length_dim < 0:
            length_dim = xs.dim() + length_dim
        ind = tuple(slice(None) if i in (0, length_dim) else None for i in range(xs.dim()))
        mask = mask[ind].expand_as(xs).to(xs.device)
    return mask


def make_non_pad_mask(lengths, xs=None, length_dim=-1, maxlen=None):
    return ~make_pad_mask(lengths, xs, length_dim, maxlen)


def _lfilter_zero_state(b, a, x):
    """direct-form IIR along the last axis, zero initial state (float64)."""
    from scipy import signal

    return signal.lfilter(b, a, x, axis=-1)


def lowpass_filter(x, fs=100, cutoff=20, N=5):
    """Zero-phase Butterworth low-pass of the log-F0 track (utils/model.py:164-196).

    For tensors the reference calls ``torchaudio.functional.filtfilt(x, a, b,
    clamp=False)`` = forward lfilter, flip, lfilter, flip with ZERO initial state
    (not scipy.signal.filtfilt's edge padding).  torchaudio is an un-vendored,
    absent dependency: that published algorithm is restated here (parity for this
    helper is therefore unpinned); numpy inputs keep
    scipy.signal.filtfilt like the reference.  Device tensors are filtered by the
    ``ptpp_filtfilt`` kernel (double precision, no host synchronisation); CPU tensors by scipy.
    """
    from scipy import signal

    nyquist = fs // 2
    b, a = signal.butter(N, [cutoff / nyquist], "lowpass")
    if x.shape[-1] <= max(len(a), len(b)) * (N // 2 + 1):
        return x  # too short to filter
    if isinstance(x, torch.Tensor):
        if x.is_cuda:  # on the device, no host round trip / synchronisation (ptpp_filtfilt)
            from .. import ops

            return ops.filtfilt(x, b, a).to(x.dtype)
        xn = x.detach().double().cpu().numpy()
        y = _lfilter_zero_state(b, a, xn)
        y = _lfilter_zero_state(b, a, y[..., ::-1])[..., ::-1]
        return torch.from_numpy(np.ascontiguousarray(y)).to(dtype=x.dtype, device=x.device)
    return signal.filtfilt(b, a, x)
