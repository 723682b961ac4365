"""Dev probe: what does torch.cuda.Event.record cost on a BUSY stream (ROCm 7 / torch 2.10), and which closing sequence of a
device-synchronised step is cheapest on the host?  A ~300 us kernel is queued, then the sequence is timed with perf_counter."""
import time, torch
pc = time.perf_counter
dev = torch.device("cuda")
a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
st = torch.cuda.current_stream(dev)


def busy():
    for _ in range(2):
        torch.mm(a, a)


def med(f, n=200):
    out = []
    for _ in range(n):
        out.append(f())
    out.sort()
    return out[len(out) // 2] * 1e6


busy(); torch.cuda.synchronize()
t0 = pc(); busy(); t_launch = pc() - t0; torch.cuda.synchronize(); t_all = pc() - t0
print("the queued work: launch %.1f us, complete after %.1f us" % (t_launch * 1e6, t_all * 1e6))
ev_t = torch.cuda.Event(enable_timing=True); ev_n = torch.cuda.Event(enable_timing=False); ev_s = torch.cuda.Event(enable_timing=True)


def rec(ev):
    def f():
        st.synchronize(); busy(); t = pc(); ev.record(st); d = pc() - t; st.synchronize(); return d
    return f


print("record on a busy stream: timing event (re-used) %.1f us, no-timing event (re-used) %.1f us, fresh timing event %.1f us" %
      (med(rec(ev_t)), med(rec(ev_n)), med(lambda: rec(torch.cuda.Event(enable_timing=True))())))


def close_a():   # record in stream order, one stream synchronisation
    st.synchronize(); ev_s.record(st); busy(); t = pc(); ev_t.record(st); st.synchronize(); e = ev_s.elapsed_time(ev_t); return pc() - t


def close_b():   # the reference: synchronise, record, synchronise
    st.synchronize(); ev_s.record(st); busy(); t = pc(); st.synchronize(); ev_t.record(st); st.synchronize(); e = ev_s.elapsed_time(ev_t); return pc() - t


def close_c():   # record in stream order, synchronise the EVENT
    st.synchronize(); ev_s.record(st); busy(); t = pc(); ev_t.record(st); ev_t.synchronize(); e = ev_s.elapsed_time(ev_t); return pc() - t


def close_d():   # no events: host clock around one synchronisation
    st.synchronize(); t0_ = pc(); busy(); t = pc(); st.synchronize(); e = pc() - t0_; return pc() - t


base = med(lambda: (st.synchronize(), busy(), pc())[2] * 0 + (lambda t: (st.synchronize(), pc() - t)[1])(pc()))
for name, f in (("a: record, stream sync", close_a), ("b: sync, record, sync (reference)", close_b), ("c: record, event sync", close_c),
                ("d: one sync, host clock", close_d)):
    print("closing sequence %-36s %.1f us from the last launch to the return" % (name, med(f)))
