"""Module-level stand-ins for the three classes ``train_base_command`` is handed, small enough to run on CPU ranks over gloo:
what the ``--devices N`` launcher test needs to observe the driver's own behaviour (spec passing, one log directory, rank
shards, sharded validation, rank-0 checkpoints) without a GPU.  Test infrastructure only."""

from __future__ import annotations

import json
import os
from pathlib import Path

import torch

from everyvoice_amd.config import HiFiGANConfig
from everyvoice_amd.dataset import ShardedSampler
from everyvoice_amd.lightning import _Module


class Outer:
    class StubConfig(HiFiGANConfig):
        """(nested on purpose: the launcher must resolve qualified names)"""


class StubData:
    def __init__(self, config, rank: int = 0, world: int = 1):
        self.config, self.rank, self.world = config, rank, world
        self.train_sampler = None

    def prepare_data(self):
        Path(self.config.training.logger.save_dir).mkdir(parents=True, exist_ok=True)
        (Path(self.config.training.logger.save_dir) / "prepared.txt").write_text(f"rank {os.environ.get('RANK', '0')}\n")

    def setup(self, stage):
        self.items = list(range(8))

    def train_dataloader(self):
        self.train_sampler = ShardedSampler(len(self.items), self.rank, self.world, shuffle=False)
        idx = list(self.train_sampler)
        return [torch.tensor(idx[i : i + 2]) for i in range(0, len(idx), 2)]

    def val_dataloader(self):
        return [torch.tensor([float(i)]) for i in range(5)]


class StubModel(_Module):
    def __init__(self, config, scale: float = 1.0, tag: str = "", process_group=None):
        super().__init__()
        self.config, self.scale, self.tag, self.process_group = config, scale, tag, process_group
        self._steps, self.seen, self.val_seen = 0, [], []
        self.trainer_ = None

    @property
    def global_step(self):
        return self._steps

    def to(self, device):
        self.device = torch.device(device)
        return self

    def training_step(self, batch, batch_idx=0):
        import torch.distributed as dist

        self.seen += batch.tolist()
        total = batch.sum().to(torch.float64).reshape(1)
        if dist.is_initialized():
            dist.all_reduce(total)  # the step's gradient exchange stands in here: every rank must reach it
        self._steps += 1
        return {"total": float(total)}

    def validation_step(self, batch, batch_idx=0):
        self.val_seen.append(batch_idx)
        return float(batch[0]) * self.scale

    def checkpoint(self):
        return {"global_step": self._steps, "epoch": self.current_epoch, "state_dict": {}}

    def on_save_checkpoint(self, ckpt):
        ckpt["model_info"] = {"name": "StubModel", "version": "1.0"}
        rank = int(os.environ.get("RANK", "0"))
        lg = self.config.training.logger
        out = Path(lg.save_dir) / f"rank{rank}.json"
        out.write_text(json.dumps({"rank": rank, "world": int(os.environ.get("WORLD_SIZE", "1")), "scale": self.scale, "tag": self.tag, "seen": self.seen,
                                   "val_seen": self.val_seen, "monitor": self.logged, "sub_dir": lg.sub_dir, "process_group": self.process_group is not None}))

