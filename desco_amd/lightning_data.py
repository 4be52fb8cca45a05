"""LightningDataLoader (subgraph_counting/lightning_data.py:59-100) without Lightning: a holder of
the three splits that returns re-iterable batch streams (``shuffle=False`` for every split, as
main.py:195)."""
from __future__ import annotations


class _Loader:
    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size

    def __iter__(self):
        return self.dataset.batches(self.batch_size)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size


class LightningDataLoader:
    def __init__(self, train_dataset=None, test_dataset=None, val_dataset=None, batch_size=64,
                 num_workers=0, shuffle=False):
        if shuffle:
            raise NotImplementedError("the reference always passes shuffle=False (main.py:195)")
        self.train_dataset, self.val_dataset, self.test_dataset = train_dataset, val_dataset, test_dataset
        self.batch_size, self.num_workers = batch_size, num_workers

    def train_dataloader(self):
        return _Loader(self.train_dataset, self.batch_size)

    def val_dataloader(self):
        return _Loader(self.val_dataset, self.batch_size)

    def test_dataloader(self):
        return _Loader(self.test_dataset, self.batch_size)
