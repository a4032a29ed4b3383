"""rankpo_amd -- MI355X-native scoring hot path of a contrastive / RankPO embedding trainer.

Importing the package loads librankpo_hip.so (built by `__graft_entry__.build()` / `make -C rankpo_amd/csrc`);
there is no CPU fallback for the hot path, so a missing library is an ImportError here, not a silent slow path.
"""
from . import _lib

_lib.load()

from .modeling import ModelForInference, ModelForTraining, ModelOutput  # noqa: E402
from .rankpo_trainer import RankPOTrainer  # noqa: E402
from .data_utils import ContrastiveDataCollatorWithPadding, RankPODataCollatorWithPadding  # noqa: E402
from . import ops  # noqa: E402

__all__ = ["ModelForTraining", "ModelForInference", "ModelOutput", "RankPOTrainer",
           "ContrastiveDataCollatorWithPadding", "RankPODataCollatorWithPadding", "ops"]
__version__ = "0.1.0"
