#!/usr/bin/env python3
"""Which CPUs this process may run on, and which are local to GPU 0 (sysfs)."""
import glob, os
print("allowed cpus:", sorted(os.sched_getaffinity(0)))
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        vendor = open(d + "/vendor").read().strip()
        if vendor != "0x1002": continue
        print(d, "->", os.path.realpath(d).split("/")[-1], "numa_node", open(d + "/numa_node").read().strip(),
              "local_cpulist", open(d + "/local_cpulist").read().strip())
    except OSError as e:
        print(d, e)
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n, open(n).read().strip())
try:
    import torch
    p = torch.cuda.get_device_properties(0)
    print("torch device 0 pci:", getattr(p, "pci_domain_id", "?"), getattr(p, "pci_bus_id", "?"), getattr(p, "pci_device_id", "?"))
except Exception as e:
    print("torch:", e)
