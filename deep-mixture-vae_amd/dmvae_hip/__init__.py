"""dmvae_hip -- MI355X-native (gfx950) DMVAE training step behind a C ABI.

Importing this package loads libdmvae_hip.so (built by ../build.py); it raises
ImportError if the library is absent.  Nothing in here computes on the CPU."""
from . import _lib
from ._lib import lib, check, DmvaeError, LIB_PATH
from .runtime import Session, StepEngine, default_session, prof_enable, prof_collect, layer_table, latent_eval
from .parallel import GradExchange, ShardedExchange, make_exchange, shard_range, epoch_plan

__all__ = ["lib", "check", "DmvaeError", "LIB_PATH", "Session", "StepEngine", "default_session",
           "prof_enable", "prof_collect", "layer_table", "GradExchange", "ShardedExchange", "make_exchange",
           "shard_range", "epoch_plan"]
