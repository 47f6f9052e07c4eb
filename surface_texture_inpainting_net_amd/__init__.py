"""MI355X-native STINet graph-convolution hot path."""
import os as _os

import torch as _torch

# ROCm 7.2 CLR: replaying a LINEAR captured HIP graph through the runtime's pre-built AQL packets ("graph packet capture")
# faults with "Write access to a read-only page" once eager work (a pinned allocation + a device-to-host copy inside a
# step) has run between two replays - reproduced with train_step.TrainStep(graph=True) at 20k vertices, gone with the
# packet path switched off (profiles/_graph_debug.py; host cost of a replay unchanged).  The flag is read when the HIP
# runtime initialises, i.e. at the first GPU call: importing this package before that is enough (TrainStep(graph=True)
# refuses to run otherwise).  Eager mode is unaffected.
GRAPH_REPLAY_SAFE = _os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0' or not _torch.cuda.is_initialized()
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

__version__ = '0.1.0'
