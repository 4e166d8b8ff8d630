"""Batch iteration with the reference loader's boundaries.

The GNN driver uses ``DataLoader(TensorDataset(train_indices, train_labels), batch_size=10000,
shuffle=False)`` (gnn/marglik_training.py:125-127): contiguous slices, last one ragged.
``TensorBatchLoader`` yields exactly those slices without per-sample collation (which costs more
than the GPU work at 10 000 samples per batch) and works on device-resident tensors.  Any other
iterable of ``(X, y)`` batches with a ``.dataset`` of known length works with ``fit`` too.
"""
from __future__ import annotations

import torch


class _Dataset:
    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n


class TensorBatchLoader:
    def __init__(self, indices: torch.Tensor, labels: torch.Tensor, batch_size: int = 10000):
        if indices.shape[0] != labels.shape[0]:
            raise ValueError("indices and labels must have the same length")
        if batch_size <= 0:
            raise ValueError("batch_size must be positive")
        self.indices, self.labels, self.batch_size = indices, labels, int(batch_size)
        self.dataset = _Dataset(indices.shape[0])

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for s in range(0, len(self.dataset), self.batch_size):
            yield self.indices[s:s + self.batch_size], self.labels[s:s + self.batch_size]


def batches_of_rank(num_batches: int, rank: int, world: int) -> list[int]:
    """Batch t belongs to rank t mod world: whole batches only."""
    return [t for t in range(num_batches) if t % world == rank]


def units_of_rank(num_batches: int, num_classes: int, rank: int, world: int) -> list[tuple[int, int, int]]:
    """Finer data-parallel decomposition of a KFAC fit: the work unit is (batch t, class column c) because
    ``B_l = sum_t sum_c g_{t,c}^T g_{t,c}`` (one reference backward pass per class column,
    curvlinops/kfac.py:653-661).  The T*C units are dealt in contiguous, equally sized runs; a rank gets
    ``[(t, class_begin, class_end), ...]`` -- whole batches where its run covers them, class ranges at the
    seams.  Batches are never split by SAMPLES (cross-sample terms inside a batch)."""
    total = num_batches * num_classes
    lo, hi = total * rank // world, total * (rank + 1) // world
    out = []
    u = lo
    while u < hi:
        t, c = divmod(u, num_classes)
        ce = min(num_classes, c + (hi - u))
        out.append((t, c, ce))
        u += ce - c
    return out
