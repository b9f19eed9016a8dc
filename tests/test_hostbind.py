"""Host-side NUMA confinement helper (gaussianprocessnode_amd/hostbind.py): parsing and the do-nothing cases (CPU only)."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    "_sgp_hostbind", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussianprocessnode_amd", "hostbind.py"))
hostbind = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(hostbind)


def test_cpulist_parsing():
    assert hostbind._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert hostbind._cpulist("") == set()
    assert hostbind._cpulist("5") == {5}


def test_bind_is_a_no_op_when_disabled_or_on_one_node(monkeypatch):
    before = os.sched_getaffinity(0)
    monkeypatch.setenv("SGP_NO_HOST_BIND", "1")
    assert hostbind.bind_to_gpu_node(0)["how"] == "none"
    monkeypatch.delenv("SGP_NO_HOST_BIND")
    monkeypatch.setattr(hostbind, "numa_nodes", lambda: {0: set(before)})
    assert hostbind.bind_to_gpu_node(0)["how"] == "none"
    assert os.sched_getaffinity(0) == before


def test_bind_picks_the_gpus_node_and_falls_back_to_the_current_one(monkeypatch):
    before = os.sched_getaffinity(0)
    cpus = sorted(before)
    if len(cpus) < 2:
        return
    half = len(cpus) // 2
    nodes = {0: set(cpus[:half]), 1: set(cpus[half:])}
    monkeypatch.setattr(hostbind, "numa_nodes", lambda: nodes)
    try:
        monkeypatch.setattr(hostbind, "visible_gpus", lambda: [{"bdf": "0000:23:00.0", "numa_node": 1, "unique_id": 1, "render_minor": 128}])
        info = hostbind.bind_to_gpu_node(0)
        assert info["how"] == "gpu" and info["node"] == 1 and os.sched_getaffinity(0) == nodes[1]
        os.sched_setaffinity(0, before)
        monkeypatch.setattr(hostbind, "visible_gpus", lambda: [])
        info = hostbind.bind_to_gpu_node(3)
        assert info["how"] == "current" and os.sched_getaffinity(0) == nodes[info["node"]]
    finally:
        os.sched_setaffinity(0, before)


def test_visible_devices_variables_select_by_index(monkeypatch):
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    # (no KFD topology in the CPU container: the list is empty and stays empty; the call must not raise)
    assert hostbind.visible_gpus() == [] or isinstance(hostbind.visible_gpus(), list)
