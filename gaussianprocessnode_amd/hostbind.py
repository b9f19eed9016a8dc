"""Keep the host side of a rank on the CPUs of ONE NUMA node -- the GPU's own when it can be told from sysfs.

Why: a sweep is ~36 kernel launches, and until the host is a sweep ahead the device waits for them.  On the two-socket
MI355X hosts an unconfined process runs its short blocks (the driver's `--steps 20`) ~6 % slower one start in four
(device sweep 231 instead of 217 us, every block of that process), confined to either node never
(profiles/r04_ab_log.txt [37]: 6/6 starts on node 0, 6/6 on node 1, 2 slow of 6 unconfined; the GPU's own node is
another 0.7 % ahead of the other).  This is what `numactl --cpunodebind` does; it is the launcher's job (bench.py calls
it first thing, before torch or HIP create a thread), never the library's: `libsgp_hip.so` does not touch affinities.

No HIP, no torch: the GPU is found through the KFD topology (the order ROCr enumerates in), the render nodes this
process may open and the *_VISIBLE_DEVICES variables; whatever cannot be resolved falls back to the node the process is
running on.
"""
from __future__ import annotations

import glob
import os
import re


def _cpulist(text: str) -> set[int]:
    out: set[int] = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def numa_nodes() -> dict[int, set[int]]:
    nodes = {}
    for p in glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"):
        try:
            nodes[int(re.search(r"node(\d+)/cpulist$", p).group(1))] = _cpulist(open(p).read())
        except (OSError, ValueError, AttributeError):
            pass
    return nodes


def visible_gpus() -> list[dict]:
    """GPUs in ROCr's enumeration order that this process can open: [{'bdf', 'numa_node', 'unique_id', 'render_minor'}]."""
    gpus = []
    paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/[0-9]*")
    for d in sorted(paths, key=lambda p: int(os.path.basename(p))):
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(d, "properties")).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) == 0:
                continue                                     # a CPU node
            minor = int(props["drm_render_minor"])
            if not os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK):
                continue                                     # (the device cgroup of a one-GPU box hides the other seven)
            dev = f"/sys/class/drm/renderD{minor}/device"
            gpus.append({"bdf": os.path.basename(os.path.realpath(dev)), "numa_node": int(open(dev + "/numa_node").read()),
                         "unique_id": int(props.get("unique_id", "0")), "render_minor": minor})
        except (OSError, ValueError, KeyError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):       # applied in this order by the stack
        val = os.environ.get(var)
        if not val:
            continue
        picked = []
        for tok in val.split(","):
            tok = tok.strip()
            if tok.isdigit() and int(tok) < len(gpus):
                picked.append(gpus[int(tok)])
            elif tok.upper().startswith("GPU-"):
                picked += [g for g in gpus if f"{g['unique_id']:016x}" == tok[4:].lower()]
            else:
                return gpus                                  # (a form this parser does not know: leave the list alone)
        gpus = picked
    return gpus


def bind_to_gpu_node(local_rank: int = 0) -> dict:
    """Confine the calling process (call it before any thread exists) to the allowed CPUs of one NUMA node; returns what was
    done: {'node', 'cpus', 'how', 'bdf'} -- 'how' is 'gpu' (the node of GPU `local_rank`), 'current' (the node this thread was
    running on) or 'none' (one node only, nothing to choose, or SGP_NO_HOST_BIND set)."""
    info = {"node": None, "cpus": None, "how": "none", "bdf": None}
    if os.environ.get("SGP_NO_HOST_BIND") or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        allowed = os.sched_getaffinity(0)
        nodes = {k: v & allowed for k, v in numa_nodes().items() if v & allowed}
        if len(nodes) < 2:
            return info
        node, how = None, "current"
        gpus = visible_gpus()
        if 0 <= local_rank < len(gpus) and gpus[local_rank]["numa_node"] in nodes:
            node, how, info["bdf"] = gpus[local_rank]["numa_node"], "gpu", gpus[local_rank]["bdf"]
        if node is None:
            cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else min(allowed)
            node = next((k for k, v in nodes.items() if cpu in v), None)
        if node is None:
            return info
        os.sched_setaffinity(0, nodes[node])
        info.update(node=node, cpus=len(nodes[node]), how=how)
    except OSError:
        pass
    return info


def rebind_all_threads(node: int) -> bool:
    """Late correction (the GPU turned out to sit on another node than `bind_to_gpu_node` assumed): every thread of the process."""
    try:
        cpus = numa_nodes()[node]
        for t in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(t), cpus)
            except OSError:
                pass
        return True
    except (OSError, KeyError):
        return False
