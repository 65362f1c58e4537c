"""Checkpoint / ensemble layer around the train step.

* ``save_generator`` / ``load_generator``: the reference's ``best_model.pth`` (GAN_DANet_train.ipynb:L275, L282, L299):
  the generator's ``state_dict`` with the reference's own keys -- a file written here loads into the reference
  module and vice versa.
* ``EarlyStopping``: the patience / min_delta rule of L271-283.
* ``save_training_state`` / ``load_training_state``: what the reference does NOT keep -- discriminator, both AdamW
  states, the LR schedulers and the epoch counter -- so that a run resumes to fp32 round-off (bit-for-bit under
  ``gd.set_deterministic(True)``: the default split-K / weight-gradient / PAM dQ sums use fp32 atomics, whose order
  varies from launch to launch).
* under data parallelism only rank 0 writes files (replicas are identical); every rank reads.
* ensemble (deep_ensemble.ipynb c0:270-337): members are independent replicas with seeds 42 + i; on an N-GPU node
  member i trains on rank i % N with NO gradient exchange (``GanTrainer(reduce_gradients=False)``), and the
  prediction mean / spread of ``compute_uncertainty`` (c0:410-448) is a reduction over the member axis.

Files are read with ``torch.load(weights_only=True)`` only.
"""
from __future__ import annotations

import os
import random
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from .parallel import rank as _rank


def _atomic_save(obj, path: str) -> None:
    tmp = f"{path}.tmp{os.getpid()}"
    torch.save(obj, tmp)
    os.replace(tmp, path)          # a crash mid-write never leaves a truncated file under the final name


def save_generator(G: nn.Module, path: str = "best_model.pth", all_ranks: bool = False) -> None:
    """rank 0 only under data parallelism (replicas are identical; ``all_ranks`` for independent ensemble members)"""
    if all_ranks or _rank() == 0:
        _atomic_save(G.state_dict(), path)


def load_generator(G: nn.Module, path: str = "best_model.pth", device=None) -> nn.Module:
    G.load_state_dict(torch.load(path, map_location=device or "cpu", weights_only=True))
    return G


class EarlyStopping:
    """L208-291: keep the best generator, stop after ``patience`` (20 in the notebook, L208) epochs without an
    improvement > ``min_delta``; ``finish`` reloads the best weights at a normal end of training too (L290, L307).
    ``all_ranks``: every process writes (independent ensemble members with their own paths)."""

    def __init__(self, patience: int = 20, min_delta: float = 0.0, path: str = "best_model.pth",
                 all_ranks: bool = False) -> None:
        self.patience, self.min_delta, self.path, self.all_ranks = patience, min_delta, path, all_ranks
        self.best_loss = float("inf")
        self.trigger_times = 0

    def _rank_uniform(self, loss: float, G: nn.Module) -> float:
        """Under data parallelism every rank holds its own SHARD's mean loss: comparing those would let best_loss /
        trigger_times drift apart, one rank reach its patience and enter ``finish``'s barrier while the others start the
        next step's collectives.  The decision is made on the mean over ranks (one scalar all-reduce per epoch), so all
        ranks save / count / stop together.  Independent members (``all_ranks``) keep their own loss."""
        import torch.distributed as dist
        if self.all_ranks or not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return float(loss)
        dev = next(G.parameters()).device
        if dist.get_backend() == "gloo":
            dev = torch.device("cpu")
        t = torch.tensor([float(loss)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.item() / dist.get_world_size()

    def step(self, avg_epoch_loss_g: float, G: nn.Module) -> bool:
        """returns True when training should stop (the best weights are then already loaded back into G); collective
        under data parallelism (call on every rank, once per epoch)"""
        avg_epoch_loss_g = self._rank_uniform(avg_epoch_loss_g, G)
        if avg_epoch_loss_g < self.best_loss - self.min_delta:
            self.best_loss = avg_epoch_loss_g
            self.trigger_times = 0
            save_generator(G, self.path, self.all_ranks)
            return False
        self.trigger_times += 1
        if self.trigger_times >= self.patience:
            self.finish(G)
            return True
        return False

    def finish(self, G: nn.Module) -> nn.Module:
        """load the best generator back (end of training, L290 / L307); under data parallelism the other ranks wait
        for rank 0's file"""
        import torch.distributed as dist
        if not self.all_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
        if os.path.exists(self.path):
            load_generator(G, self.path, next(G.parameters()).device)
        return G


def save_training_state(path: str, trainer, epoch: int, schedulers: Sequence = (), extra: Optional[Dict] = None) -> None:
    """collective under data parallelism (the optimiser state of sharded tensors is gathered): call on every rank"""
    if hasattr(trainer, "sync_params"):
        trainer.sync_params()
    state = {
        "epoch": int(epoch),
        "G": trainer.G.state_dict(),
        "D": trainer.D.state_dict(),
        "opt_g": trainer.opt_g.state_dict(),
        "opt_d": trainer.opt_d.state_dict(),
        "schedulers": [s.state_dict() for s in schedulers],
        "extra": dict(extra or {}),
    }
    if getattr(trainer, "input_attention", None) is not None:
        state["input_attention"] = trainer.input_attention.state_dict()
    if _rank() == 0 or not getattr(trainer, "_reduce", True):      # replicas are identical: one writer
        _atomic_save(state, path)


def load_training_state(path: str, trainer, schedulers: Sequence = ()) -> Dict:
    dev = next(trainer.G.parameters()).device
    state = torch.load(path, map_location=dev, weights_only=True)
    trainer.G.load_state_dict(state["G"])
    trainer.D.load_state_dict(state["D"])
    trainer.opt_g.load_state_dict(state["opt_g"])
    trainer.opt_d.load_state_dict(state["opt_d"])
    if "input_attention" in state and getattr(trainer, "input_attention", None) is not None:
        trainer.input_attention.load_state_dict(state["input_attention"])
    for s, sd in zip(schedulers, state["schedulers"]):
        s.load_state_dict(sd)
    return {"epoch": state["epoch"], "extra": state["extra"]}


# ---- ensemble ------------------------------------------------------------------------------------------------
def member_seed(i: int) -> int:
    return 42 + i                                                    # deep_ensemble.ipynb c0:284


def set_seed(seed: int) -> None:                                     # deep_ensemble.ipynb c0:286-292
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def members_of_rank(num_ensemble: int, rank: int, world: int) -> List[int]:
    """independent members spread over the ranks of a node: member i -> rank i % world"""
    return [i for i in range(num_ensemble) if i % world == rank]


def member_path(ensemble_dir: str, i: int) -> str:
    return os.path.join(ensemble_dir, f"best_model_member_{i + 1}.pth")   # c0:310 (1-based file names)


@torch.no_grad()
def predict_ensemble(models: Iterable[nn.Module], x: torch.Tensor):
    """mean and standard deviation over the members' predictions (c0:339-448: ``np.mean`` / ``np.std`` over the member
    axis, population std).  The member forwards are the HIP generator; the reduction over a handful of members is
    plain tensor arithmetic."""
    preds = torch.stack([m(x) for m in models], 0)
    return preds.mean(0), preds.std(0, unbiased=False)
