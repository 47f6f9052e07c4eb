"""MI355X-native STINet graph-convolution hot path."""
import os as _os

import torch as _torch

# ROCm 7.2 CLR: replaying a LINEAR captured HIP graph through the runtime's pre-built AQL packets ("graph packet capture")
# faults with "Write access to a read-only page" once eager work (a pinned allocation + a device-to-host copy inside a
# step) has run between two replays - reproduced with train_step.TrainStep(graph=True) at 20k vertices, gone with the
# packet path switched off (profiles/probes/graph_debug.py; host cost of a replay unchanged).  The flag is read when the HIP
# runtime initialises, i.e. at the first GPU call, so it has to be in the environment before that.  It is an OPT-IN: only
# a process that asks for graph replay gets its environment changed - enable_graph_replay() before the first GPU call, or
# STIN_GRAPH_REPLAY=1 exported when the package is imported.  Eager mode never needs it and is left alone.
_FLAG = 'DEBUG_CLR_GRAPH_PACKET_CAPTURE'
_state = {'safe': _os.environ.get(_FLAG) == '0'}


def enable_graph_replay():
    """Make train_step.TrainStep(graph=True) usable in this process: sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, which only works
    while the HIP runtime is not initialised yet.  -> True when graph replay is safe from here on."""
    if _os.environ.get(_FLAG) == '0' and _state['safe']:
        return True
    if _torch.cuda.is_initialized():
        return False
    # torch.cuda.is_initialized() only knows about torch's own lazy init: under a profiler whose preloaded library has already
    # brought the HIP runtime up (rocprofv3 with --pmc does), setting the flag now may be too late to take effect - refuse
    # rather than report the replay path safe
    pre = _os.environ.get('LD_PRELOAD', '')
    if any(k in pre for k in ('rocprofiler', 'roctracer', 'rocprof')) or any(k.startswith(('ROCPROFILER_', 'ROCPROF_', 'ROCP_TOOL')) for k in _os.environ):
        if _os.environ.get(_FLAG) != '0':
            return False
    _os.environ[_FLAG] = '0'
    _state['safe'] = True
    return True


def graph_replay_safe():
    return _state['safe'] and _os.environ.get(_FLAG) == '0'


if _os.environ.get('STIN_GRAPH_REPLAY', '0') == '1':
    enable_graph_replay()

__version__ = '0.1.0'
