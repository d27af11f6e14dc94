"""oryon_amd: the MI355X-native Oryon match -> lift -> registration path."""
import os
import warnings

__all__ = ["configure", "hw_queues_ok"]

_MIN_HW_QUEUES = 6


def hw_queues_ok() -> bool:
    """True when the process environment gives the HIP runtime enough hardware queues for the step engine's streams."""
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= _MIN_HW_QUEUES
    except ValueError:
        return False


def configure(hw_queues: int = 8) -> bool:
    """Process-level settings the step engine (csrc/engine.hip) wants, made EXPLICITLY by the host program - importing the package
    changes nothing (round 3 set the variable at import time, which silently did nothing in a host that had touched HIP already).

    The engine drives four HIP streams of its own (gather, match, two registration streams) next to the caller's.  The HIP runtime
    multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4): with five streams two share a queue, and a
    registration stream that shares the caller's queue runs its kernels behind the caller's event waits - the two registration streams
    serialise (measured: 5.0 instead of 4.7 ms per cfg2 step).  The runtime reads the variable once, at its first HIP call, so this
    must run before anything initialises HIP (torch.cuda.init, the first CUDA tensor, the first call into liboryon_hip.so).

    Returns True when the setting is (already, or now) in effect for a runtime that has not started yet; warns and returns False when
    HIP is already initialised with fewer queues - results are unaffected, the pipelined step is ~5 % slower."""
    if hw_queues_ok():
        return True
    started = False
    try:
        import torch
        started = torch.cuda.is_initialized()
    except Exception:
        pass
    if started:
        warnings.warn(f"oryon_amd.configure(): HIP is already initialised with GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '4 (default)')}; "
                      f"the step engine's registration streams will share hardware queues (results unchanged, ~5 % slower). "
                      f"Call oryon_amd.configure() - or export GPU_MAX_HW_QUEUES={hw_queues} - before the first HIP call.", RuntimeWarning, stacklevel=2)
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(hw_queues)
    return True
