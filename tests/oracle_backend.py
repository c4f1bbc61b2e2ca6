"""A StreamProcessor backend that computes with the CPU oracle -- TEST ONLY (lets the sharding
and gather plumbing of passiveradar_amd.stream run under gloo without a GPU)."""
import numpy as np
import torch

from oracle import np_oracle as O


class OracleBackend:
    def __init__(self, cpi_samples, R, F, fs, bins=(0, 1, -1, 2, -2)):
        self.cpi, self.C, self.R, self.F, self.fs, self.bins = cpi_samples, cpi_samples // 2, R, F, fs, bins
        self.device = "cpu"
        from scipy.signal import get_window
        self.window = get_window(("kaiser", 5.0), cpi_samples)

    def padded(self, chunks):
        x = np.asarray(chunks, dtype=np.complex64)
        z = np.zeros(self.C // 2, np.complex64)
        return np.concatenate((z, x, z))

    def clean(self, ref_pad, srv_pad, nlocal):
        C, h = self.C, self.C // 2
        out = np.zeros_like(srv_pad)
        for c in range(nlocal):
            sl = slice(h + c * C, h + (c + 1) * C)
            out[sl] = O.LS_Filter_Multiple(ref_pad[sl], srv_pad[sl], self.R, self.fs, list(self.bins))
        return out

    def frames(self, ref_pad, clean_pad, first, nframes):
        fr = [O.fast_xambg(ref_pad[first + i * self.C:first + i * self.C + self.cpi],
                           clean_pad[first + i * self.C:first + i * self.C + self.cpi],
                           self.R, self.F, self.cpi, self.window)[:, :, 0] for i in range(nframes)]
        return torch.from_numpy(np.stack(fr).astype(np.complex64))
