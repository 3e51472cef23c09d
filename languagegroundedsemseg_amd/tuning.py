"""ONE table of the host-side (Python) tuning / debugging knobs.  The engine-side knobs live in the C table of
csrc/lgs_tuning.hip (engine.tuning_table(), engine.tuning_set()); `python -m languagegroundedsemseg_amd.tuning` prints both.

A knob's value is the environment variable LGS_<NAME> when set at import time, else its default.  None of them changes results
(the A/B ones are bit-identical paths, proven by the tests named in their descriptions)."""
import os

HOST_KNOBS = {
    # name: (default, parser, doc)
    "WGRAD_INLINE_BELOW": (60000, int, "batches below this many input voxels run their weight gradients on the compute stream (no fork / "
                                       "join events); larger ones use the side stream.  Was 400000 while a one-scene step was host-bound; "
                                       "with lgs_block_backward forking inside the engine call one 145 k-voxel scene per step runs "
                                       "9.6 - 9.9 ms on the side stream against 10.1 - 10.2 inline (backward 5.3 vs 6.6 ms of stream time)"),
    "DEFER": (1, int, "0 = every MinkowskiEngine call executes immediately (the reference's call sequence runs unfused: separate norm, "
                      "ReLU, add and concat-copy kernels) instead of being recorded and run fused when a value is first read "
                      "(me/deferred.py; same results: tests/test_gpu_reference_calls.py)"),
    "DEFER_INCREMENTAL": (1, int, "1 = recorded calls execute as soon as no later call can change them (a norm once its result is "
                                  "consumed, a convolution once it is known whether it opens a residual block): the GPU works on a layer "
                                  "while the host records the next; 0 = nothing executes until a value is read (one scene per step: "
                                  "the whole forward is then recorded before the first launch, +1 - 2 ms of latency)"),
    "BLOCK_FUSED": (1, int, "0 = BasicBlocks run module by module instead of as one autograd node "
                            "(bit-identical: test_block_fast_path_is_the_op_by_op_path)"),
    "BLOCK_C": (1, int, "0 = a BasicBlock is enqueued call by call instead of through lgs_block_forward / lgs_block_backward "
                        "(bit-identical: test_c_side_block_equals_the_call_by_call_block)"),
    "PACK_CACHE": (1, int, "0 = re-pack the MFMA weight image on every conv call instead of once per optimiser step "
                           "(bit-identical: test_packed_weight_cache_never_serves_stale_weights)"),
    "ZERO_COPY_CAT": (1, int, "0 = ME.cat copies instead of both norms writing into the concat buffer "
                              "(bit-identical: test_zero_copy_cat_equals_the_copying_cat)"),
    "WIDE_WGRAD_INLINE": (0, int, "1 = weight gradients of >= 256 x 256-channel layers on big maps run on the compute stream instead of the side "
                                  "stream (A/B, round 4: CLIP step 166.8 vs 164.0 ms -- every wide kernel then runs at its stand-alone time, "
                                  "e.g. the 1x1 512->544 dgrad 1.37 instead of 11.6 ms, but the step is the sum of its kernels either way)"),
    "CONV_BN_STATS": ("", str, "'1' / 'big' = BatchNorm statistics from the conv epilogue (measured slower: 31.5 vs 30.9 ms; off)"),
    "DBG_WGRAD": ("", str, "'skip' / 'inline': step-time attribution experiments only ('skip' produces no weight gradients)"),
    "SYNCBN_ENGINE_COMM": (-1, int, "MinkowskiSyncBatchNorm's collectives issued by the engine on its own RCCL communicator, on the compute "
                                    "stream (csrc/lgs_comm.hip): 1 = always, 0 = never (torch.distributed collectives between the split "
                                    "kernels), -1 = auto: only in a world of ONE rank (two communicators in flight next to each other have "
                                    "never been executed with a peer on this build's one-GPU boxes; ddp.EngineComm)"),
    "SYNCBN_IPC": (0, int, "1 = MinkowskiSyncBatchNorm exchanges its per-layer records through device-side mailboxes (peers' buffers mapped with "
                           "hipIpc, ONE kernel per exchange that writes to every rank and spins on arrival flags; csrc/lgs_comm.hip) instead of "
                           "RCCL collectives.  Off: visibility of peer stores to a spinning kernel across xGMI has never been exercised (one GPU "
                           "per box); the logic is tested with two processes on one GPU (tests/test_gpu_syncbn.py)"),
    "SYNCBN_OWN_GROUP": (1, int, "MinkowskiSyncBatchNorm's torch.distributed collectives (the N > 1 default) run on a process group of their own -- "
                                 "a second communicator and stream -- instead of the default group, whose one RCCL stream also carries the 32 MB "
                                 "gradient-bucket all-reduces: behind one of those a 800-byte statistics exchange on the backward pass's critical path "
                                 "waits ~0.2 - 0.3 ms, up to six times per step.  1 = for the default group of an `nccl` world of >= 2 ranks, "
                                 "2 = for any backend and world size (tests), 0 = off"),
    "SET_HW_QUEUES": (0, int, "1 = importing the package sets GPU_MAX_HW_QUEUES=8 before the HIP runtime starts (see configure_hw_queues)"),
}


def host(name):
    default, parse, _ = HOST_KNOBS[name]
    v = os.environ.get("LGS_" + name)
    if v is None:
        return default
    try:
        return parse(v)
    except ValueError:
        raise ValueError("LGS_%s=%r: expected %s" % (name, v, parse.__name__))


def configure_hw_queues(n=8):
    """The engine drives FOUR HIP streams per device (compute, weight gradients, coordinate / kernel maps, input staging) next to
    torch's and RCCL's.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with a fifth
    stream two of them share a queue and run IN ORDER -- measured: the map stream landed behind the compute stream, so every
    `SparseTensor(...)` (its insert returns a count to the host) blocked the host until the GPU had finished the previous step
    (one 145 k-voxel scene per step: 11.5 vs 10.4 ms).  The variable is read when the HIP runtime initialises, so this must run
    before the process's first device call; an explicit user setting wins.  It is a PROCESS-WIDE setting and therefore the
    application's call (bench.py and tests/conftest.py make it), not a side effect of importing a library: several processes
    time-sharing ONE GPU should keep the runtime's default (2 x 8 queues oversubscribe the device: 136 vs 520 ms per step)."""
    import warnings
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return int(os.environ["GPU_MAX_HW_QUEUES"])
    try:
        import torch
        if torch.cuda.is_initialized():
            warnings.warn("configure_hw_queues(): the HIP runtime is already initialised; GPU_MAX_HW_QUEUES has no effect now")
    except ImportError:
        pass
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return n


def describe():
    rows = [("host", k, str(d), str(host(k)), doc) for k, (d, _, doc) in HOST_KNOBS.items()]
    try:
        from . import engine
        rows += [("engine", n, str(d), str(v), doc) for n, d, v, doc in engine.tuning_table()]
    except Exception as e:      # library not built
        rows.append(("engine", "(liblgs_engine.so not built: %s)" % e, "", "", ""))
    return rows


if __name__ == "__main__":
    for side, name, d, v, doc in describe():
        print("%-6s LGS_%-20s default %-8s now %-8s %s" % (side, name, d, v, doc))
