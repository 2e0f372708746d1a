"""Host-CPU placement of one-rank-per-GPU processes (no torch import, nothing here touches the GPU)."""
import os


def _physical_cores(avail):
    """the CPUs of `avail` grouped by physical core (SMT siblings together, from sysfs), cores in ascending order of their first CPU; on
    Linux boxes numbered socket by socket that order is also socket-major, so equal slices of it keep a rank's CPUs on one socket"""
    seen, cores = set(), []
    aset = set(avail)
    for c in avail:
        if c in seen:
            continue
        sib = [c]
        try:
            with open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c) as f:
                txt = f.read().strip()
            sib = []
            for part in txt.split(','):
                a, _, b = part.partition('-')
                sib.extend(range(int(a), int(b or a) + 1))
            sib = sorted(x for x in sib if x in aset) or [c]
        except (OSError, ValueError):
            sib = [c]
        seen.update(sib)
        cores.append(sib)
    return cores


def pin_rank_to_cpus(local_rank, local_world):
    """One rank per GPU means N Python hosts on one box: each rank is confined to its own slice of the CPUs this process may run on
    (whole physical cores, equal shares, neighbouring cores; in-process os.sched_setaffinity BEFORE anything touches the GPU -- never
    taskset / numactl, which would be an exec hop), so the enqueueing threads of different ranks never migrate onto each other or share a
    core's SMT siblings.  PCP_BENCH_AFFINITY=0 (or PCP_AFFINITY=0) leaves the mask alone.  Returns the CPUs this rank runs on (None where the platform has no
    affinity call)."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if not (0 <= local_rank < local_world):          # a launcher that did not say how many ranks share this box: leave the mask alone
        return avail
    if local_world <= 1 or os.environ.get('PCP_BENCH_AFFINITY', os.environ.get('PCP_AFFINITY', '1')) == '0' or len(avail) < local_world:
        return avail
    cores = _physical_cores(avail)
    if len(cores) >= local_world:
        per = len(cores) // local_world
        mine = sorted(c for core in cores[local_rank * per:(local_rank + 1) * per] for c in core)
    else:                                           # fewer cores than ranks (SMT siblings must be split): plain CPU slices
        per = len(avail) // local_world
        mine = avail[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


def cpu_list(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    if not cpus:
        return ''
    out, a, b = [], cpus[0], cpus[0]
    for c in cpus[1:]:
        if c == b + 1:
            b = c
            continue
        out.append('%d-%d' % (a, b) if b > a else '%d' % a)
        a = b = c
    out.append('%d-%d' % (a, b) if b > a else '%d' % a)
    return ','.join(out)
