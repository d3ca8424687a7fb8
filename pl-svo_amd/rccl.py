"""RCCL communicators for plsvo_gather_poses (the C ABI's pose all-gather, include/plsvo_hip.h): created straight from librccl
through ctypes, the way a C++ host of the library would, with torch.distributed used only as the rendezvous that hands rank 0's
unique id to the other ranks.  One process per GPU; the communicator is bound to the current device.

The library is opened by SONAME so that this module, libplsvo_hip.so and torch all talk to the SAME loaded instance (torch ships
its own librccl.so.1 and loads it first; a communicator made by a second copy would be rejected by the first)."""
import ctypes as C

_rccl = None


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def lib():
    global _rccl
    if _rccl is None:
        try:
            _rccl = C.CDLL("librccl.so.1")
        except OSError:
            _rccl = C.CDLL("/opt/rocm/lib/librccl.so")
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        _rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        _rccl.ncclGetErrorString.restype = C.c_char_p
        _rccl.ncclGetErrorString.argtypes = [C.c_int]
    return _rccl


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().ncclGetErrorString(rc).decode()}")


def unique_id():
    """128 opaque bytes identifying a communicator (rank 0 creates them, every rank passes them to comm_init)"""
    uid = UniqueId()
    _chk(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return C.string_at(C.addressof(uid), 128)


def comm_init(world, rank, uid_bytes):
    """ncclCommInitRank on the current device; returns the ncclComm_t as an integer handle"""
    if len(uid_bytes) != 128:
        raise ValueError("an RCCL unique id is 128 bytes")
    uid = UniqueId()
    C.memmove(C.addressof(uid), uid_bytes, 128)
    comm = C.c_void_p()
    _chk(lib().ncclCommInitRank(C.byref(comm), int(world), uid, int(rank)), "ncclCommInitRank")
    return comm.value


def comm_destroy(comm):
    if comm:
        lib().ncclCommDestroy(C.c_void_p(comm))


def exchange_unique_id(make_id=None):
    """Every rank of the initialised torch.distributed process group returns the SAME 128 bytes: rank 0 creates them (make_id, default
    unique_id()), the process group carries them to the others.  Without a process group: this process's own id."""
    import torch.distributed as dist
    make_id = make_id or unique_id
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return make_id()
    box = [make_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = box[0]
    if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
        raise RuntimeError("the rendezvous did not deliver a 128-byte RCCL unique id")
    return bytes(uid)


def comm_over_process_group():
    """One communicator spanning the initialised torch.distributed process group (a single-rank one without it): rank 0's unique
    id travels through the process group, then every rank joins."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return comm_init(1, 0, unique_id())
    return comm_init(dist.get_world_size(), dist.get_rank(), exchange_unique_id())
