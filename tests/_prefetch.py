"""Background regeneration of the seed-defined inputs of the real-dimension checks.

Those fixtures keep seeds, not 0.5-1.7 GB state dicts: each module regenerates its pretrained / fine-tuned parameters on the host with
torch's CPU generator (single-threaded normal fills: 20-50 s per module, ~3.5 of the GPU suite's 7 minutes, with the GPU idle).  A test
module registers the host-only part of its fixture as a builder; when the collection is final the builders of the SELECTED tests start on
a small thread pool in the order their tests will run, and the fixture picks the result up (``prefetched``) -- so the next module's inputs
are drawn while the current module's kernels run.  Same generators, same seeds, same checksums: nothing about what is tested changes.
MERGEREC_PREFETCH=0 builds inline as before; without a GPU (or with the heavy set switched off) nothing starts early."""
import os

_PREFETCH = {}


def register_prefetch(name, build, match):
    """build(): host-only, thread-safe; match: substrings that a selected test's node id must ALL contain for the builder to start early"""
    _PREFETCH[name] = dict(build=build, match=tuple(match), future=None)


def prefetched(name):
    ent = _PREFETCH[name]
    fut, ent["future"] = ent["future"], None
    return fut.result() if fut is not None else ent["build"]()


def start_prefetch(session):
    if os.environ.get("MERGEREC_PREFETCH", "1") != "1" or not _PREFETCH or session.config.getoption("collectonly", False):
        return
    heavy_off = os.environ.get("MERGEREC_HEAVY_TESTS", "") == "0"
    order = []
    for name, ent in _PREFETCH.items():
        hits = [i for i, it in enumerate(session.items) if all(m in it.nodeid for m in ent["match"])
                and not (heavy_off and it.get_closest_marker("heavy") is not None)]
        if hits:
            order.append((hits[0], name))
    if not order:
        return
    import torch

    if not torch.cuda.is_available():
        return
    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(max_workers=int(os.environ.get("MERGEREC_PREFETCH_WORKERS", "3")), thread_name_prefix="prefetch")
    for _, name in sorted(order):
        _PREFETCH[name]["future"] = pool.submit(_PREFETCH[name]["build"])
    session.config._mergerec_prefetch_pool = pool  # kept alive for the session; idle threads end with the process


def seeded_state_dicts(shapes, key_order, seed_pre, std_pre, pre_checksum, ft_seeds, ft_std, ft_checksum=None):
    """(pretrained, [fine-tuned ...]) regenerated from a fixture's seeds and checked against its checksums (the generator scripts under
    oracle/ drew them the same way: oracle.ref_cpu.random_state_dict / perturbed_state_dict along the reference wrapper's key order)"""
    from collections import OrderedDict

    from oracle import ref_cpu as O

    pre0 = O.random_state_dict(shapes, seed=seed_pre, std=std_pre)
    pre = OrderedDict((k, pre0[k]) for k in key_order)
    fsum = lambda sds: float(sum(v.double().sum() for sd in sds for v in sd.values() if v.is_floating_point()))
    assert abs(fsum([pre]) - pre_checksum) < 1e-6 * abs(pre_checksum) + 1e-9
    fts = [O.perturbed_state_dict(pre, seed=s, std=ft_std) for s in ft_seeds]
    if ft_checksum is not None:
        assert abs(fsum(fts) - ft_checksum) < 1e-6 * abs(ft_checksum) + 1e-6
    return pre, fts


def stop_prefetch(session):
    """session end: builders that have not started are dropped (a run that stopped early must not keep drawing gigabytes nobody reads)"""
    pool = getattr(session.config, "_mergerec_prefetch_pool", None)
    if pool is not None:
        pool.shutdown(wait=False, cancel_futures=True)
