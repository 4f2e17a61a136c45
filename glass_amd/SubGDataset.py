"""Dataset / loader classes with the reference's interface (/root/reference/impl/SubGDataset.py).

A loader yields whole-graph tensors plus the padded node matrix and targets of a batch of
subgraphs: (x, edge_index, edge_attr, pos[perm], [z,] y[perm]).  Data-parallel TRAINING loaders are
built with shard=True: rank r keeps perm[r::world] of every batch, so ranks label and
pool disjoint subgraphs of the same replicated graph.  Loaders without the flag (validation / test) yield
the full batches on every rank, so that scores, scheduler steps and early-stop decisions are identical
everywhere."""
import torch

from . import dist as gdist


class GDataset:
    """x: node features; pos: padded [n_subgraphs, Smax] node matrix (-1 pad); y: targets."""
    def __init__(self, x, edge_index, edge_attr, pos, y):
        self.x, self.edge_index, self.edge_attr, self.pos, self.y = x, edge_index, edge_attr, pos, y
        self.num_nodes = x.shape[0]

    def __len__(self):
        return self.pos.shape[0]

    def __getitem__(self, idx):
        return self.pos[idx], self.y[idx]

    def to(self, device):
        for name in ("x", "edge_index", "edge_attr", "pos", "y"):
            setattr(self, name, getattr(self, name).to(device))
        return self


class GDataloader:
    """Iterates index batches over the subgraphs (shuffle / drop_last as torch's DataLoader) and
    returns the tuple the training loop expects."""
    def __init__(self, Gdataset, batch_size=64, shuffle=True, drop_last=False, shard=False):
        self.Gdataset, self.batch_size, self.shuffle, self.drop_last = Gdataset, batch_size, shuffle, drop_last
        self.shard = bool(shard)  # data-parallel training loader: this rank's slice of every batch
        self.generator = None  # optional torch.Generator (CPU) for reproducible shuffles

    def get_x(self):
        return self.Gdataset.x

    def get_ei(self):
        return self.Gdataset.edge_index

    def get_ea(self):
        return self.Gdataset.edge_attr

    def get_pos(self):
        return self.Gdataset.pos

    def get_y(self):
        return self.Gdataset.y

    def __len__(self):
        n = len(self.Gdataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _batches(self):
        n = len(self.Gdataset)
        order = torch.randperm(n, generator=self.generator) if self.shuffle else torch.arange(n)
        if gdist.is_distributed():
            order = gdist.broadcast_cpu(order)  # identical permutation on every rank
        order = order.to(self.Gdataset.pos.device)
        stop = n - n % self.batch_size if self.drop_last else n
        sharded = self.shard and gdist.is_distributed()
        for s in range(0, stop, self.batch_size):
            batch = order[s:s + self.batch_size]
            if sharded:
                if batch.shape[0] < gdist.world_size():
                    continue  # a tail smaller than the world would leave a rank without subgraphs: skipped on EVERY rank
                batch = gdist.shard(batch)
            yield batch

    def __iter__(self):
        self.iter = self._batches()
        return self

    def _select(self, perm):
        return self.get_pos()[perm], self.get_y()[perm]

    def __next__(self):
        pos, y = self._select(next(self.iter))
        return self.get_x(), self.get_ei(), self.get_ea(), pos, y


class ZGDataloader(GDataloader):
    """Adds the node labels z = z_fn(x, pos) of the batch before the targets."""
    def __init__(self, Gdataset, batch_size=64, shuffle=True, drop_last=False,
                 z_fn=lambda x, y: torch.zeros((x.shape[0], x.shape[1]), dtype=torch.int64), shard=False):
        super().__init__(Gdataset, batch_size, shuffle, drop_last, shard)
        self.z_fn = z_fn

    def __next__(self):
        pos, y = self._select(next(self.iter))
        return self.get_x(), self.get_ei(), self.get_ea(), pos, self.z_fn(self.get_x(), pos), y
