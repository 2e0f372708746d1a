"""What a torch.cuda.Event pair adds around one launch (diagnostic for bench.py's per-launch timings): the elapsed time of an empty
bracket and of a bracket around a one-element fill, medians over 200 trials."""
import numpy as np
import torch

d = torch.device('cuda:0')
x = torch.empty(1, device=d)
big = torch.empty(64 << 20, device=d)
s = torch.cuda.current_stream()


def bracket(fn, trials=200, busy=False):
    out = []
    for _ in range(trials):
        if busy:
            big.zero_()                                 # the queue is not empty when the bracket is enqueued (as inside a step)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        fn()
        e1.record(s)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(out)), float(np.percentile(out, 10)), float(np.percentile(out, 90))


for busy in (False, True):
    print('queue %s: empty bracket %.2f us (p10 %.2f p90 %.2f) | one-element fill %.2f us (p10 %.2f p90 %.2f)'
          % ((('busy' if busy else 'idle'),) + bracket(lambda: None, busy=busy) + bracket(lambda: x.zero_(), busy=busy)))
