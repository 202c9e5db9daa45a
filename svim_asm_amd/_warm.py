"""Earliest possible start of the HIP runtime in a fresh `svim-asm` process.

Creating the first context (hipInit, device enumeration, stream) takes 150-350 ms and nothing of STEP 1
needs the device before the BAM headers, indices and records have been walked — so bin/svim-asm starts it
on a thread BEFORE numpy and the package are imported (this module needs ctypes only), and
`_lib.Context` adopts the context the thread made.  One process per GPU: under a launcher (LOCAL_RANK set)
the process is restricted to its own device first, which also keeps the runtime from initialising the other
seven."""
import ctypes
import os
import threading

from svim_asm_amd import _timeline

_state = {"thread": None, "device": None, "handle": None, "lib": None}


def restrict_to_local_rank():
    """Multi-GPU launch: show this process only the device of its LOCAL_RANK (which then is device 0).  Returns
    the device index to use from now on, or None outside a multi-rank launch.  A caller that has restricted the
    visible devices itself is left alone (LOCAL_RANK then indexes what it left visible)."""
    if "restricted" in _state:
        return _state["restricted"]
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or os.environ.get("LOCAL_RANK") is None:
        return None
    local = os.environ["LOCAL_RANK"]
    if "HIP_VISIBLE_DEVICES" in os.environ or "ROCR_VISIBLE_DEVICES" in os.environ:
        _state["restricted"] = int(local)
    else:
        os.environ["HIP_VISIBLE_DEVICES"] = local
        _state["restricted"] = 0
    return _state["restricted"]


def restrict_to_device(device):
    """Single-process run (no launcher): show the process only the device it was asked to use, which then is device 0
    — on a node with eight GPUs the HIP runtime otherwise brings up all of them before the first kernel of a command
    that lives for half a second.  Returns the device index to use from now on.  Left alone: a multi-rank launch
    (restrict_to_local_rank); a list of visible devices the caller set is narrowed to its entry, not overridden."""
    if "restricted" in _state:
        return _state["restricted"]
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("LOCAL_RANK") is not None:
        return device
    listed = os.environ.get("HIP_VISIBLE_DEVICES")
    if listed is None and os.environ.get("CUDA_VISIBLE_DEVICES") is not None:
        # the HIP runtime honours this name too: treat the caller's list the same way
        listed = os.environ["CUDA_VISIBLE_DEVICES"]
    if listed is not None:
        # a list the caller set: device d is its d-th entry; narrow the list to that entry (one entry: nothing to do)
        entries = [e for e in listed.split(",") if e.strip()]
        if len(entries) <= 1 or int(device) >= len(entries):
            return device
        os.environ["HIP_VISIBLE_DEVICES"] = entries[int(device)].strip()
    else:
        os.environ["HIP_VISIBLE_DEVICES"] = str(int(device))  # (indexes what ROCR_VISIBLE_DEVICES leaves visible, if set)
    _state["restricted"] = 0
    _state["physical"] = int(device)
    return 0


def logical_device(device):
    """The index under which the device the user named is visible after restrict_to_device.  A process that was
    narrowed to one device and is then asked for another cannot serve the request: that is an error here, not a
    failed context creation later."""
    if "physical" in _state and "restricted" in _state:
        if _state["physical"] != int(device):
            raise RuntimeError("this process was restricted to device %d before the HIP runtime started and cannot use "
                               "device %d (bin/svim-asm derives the device from --device on the command line)"
                               % (_state["physical"], int(device)))
        return _state["restricted"]
    return device


def start(device, lib_path):
    """Begin creating the context of `device` in the background (no-op when already started)."""
    if _state["thread"] is not None or not os.path.exists(lib_path):
        return

    def create():
        try:
            lib = ctypes.CDLL(lib_path)
            lib.svx_ctx_create.restype = ctypes.c_int
            lib.svx_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
            h = ctypes.c_void_p()
            _timeline.mark("libsvx loaded")
            if lib.svx_ctx_create(int(device), ctypes.byref(h)) == 0:
                _state["handle"], _state["lib"] = h, lib
            _timeline.mark("device context created")
        except Exception:  # noqa: BLE001 — the regular path reports what is wrong
            pass
    _state["device"] = int(device)
    _state["thread"] = threading.Thread(target=create, daemon=True)
    _state["thread"].start()


def take(device):
    """The context handle made for `device` by start(), once (None if there is none)."""
    th = _state["thread"]
    if th is None:
        return None
    th.join()
    h, _state["handle"] = _state["handle"], None
    if h is not None and _state["device"] != int(device):
        # made for another device than the one that is wanted after all: give it back
        _state["lib"].svx_ctx_destroy.argtypes = [ctypes.c_void_p]
        _state["lib"].svx_ctx_destroy.restype = None
        _state["lib"].svx_ctx_destroy(h)
        return None
    return h
