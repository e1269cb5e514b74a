"""Experiment behind the overlapped data-parallel step: does an EXTERNAL event recorded by a node in the middle of a
replayed hipGraph release a side stream that waits on it — (a) not before the node's dependencies ran, (b) before the rest of the
graph has finished?  Prints the two orderings for a few replays.
    python tools/ext_event_probe.py"""
import torch

dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
c = torch.empty_like(a)
buf = torch.zeros(1, device=dev)
snap = torch.zeros(1, device=dev)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")      # torch refuses external events on ROCm ("External events are disallowed in rocm"): straight to HIP
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
ev = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(ev), 2) == 0      # hipEventDisableTiming
side = torch.cuda.Stream(device=dev)
cap = torch.cuda.Stream(device=dev)


def chain(n):
    for _ in range(n):
        torch.mm(a, b, out=c)


with torch.cuda.stream(cap):
    chain(2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        chain(20)
        buf.add_(1)
        if True:
            # hipEventRecordWithFlags(ev, stream, hipEventRecordExternal) returns hipErrorInvalidValue on ROCm 7.2 (and leaves a sticky last
            # error that torch's next launch check reports) — the explicit route: an event-record node behind the stream's current dependencies
            st, gid, graph, deps, nd = ctypes.c_int(), ctypes.c_ulonglong(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t()
            rc = hip.hipStreamGetCaptureInfo_v2(ctypes.c_void_p(cap.cuda_stream), ctypes.byref(st), ctypes.byref(gid), ctypes.byref(graph), ctypes.byref(deps), ctypes.byref(nd))
            print("capture info ->", rc, st.value, nd.value)
            node = ctypes.c_void_p()
            rc = hip.hipGraphAddEventRecordNode(ctypes.byref(node), graph, deps, nd, ev)
            print("hipGraphAddEventRecordNode ->", rc)
            rc = hip.hipStreamUpdateCaptureDependencies(ctypes.c_void_p(cap.cuda_stream), ctypes.byref(node), ctypes.c_size_t(1), 1)     # hipStreamSetCaptureDependencies
            print("hipStreamUpdateCaptureDependencies ->", rc)
        chain(60)
torch.cuda.synchronize()
ok = True
for i in range(5):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(cap):
        e0.record(cap)
        g.replay()
        e2.record(cap)
    with torch.cuda.stream(side):
        assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), ev, 0) == 0
        snap.copy_(buf)
        e1.record(side)
    torch.cuda.synchronize()
    t_side, t_all = e0.elapsed_time(e1), e0.elapsed_time(e2)
    good = snap.item() == i + 1 and t_side < 0.6 * t_all
    ok &= good
    print(f"replay {i}: snapshot {snap.item():.0f} (want {i + 1}), side released after {t_side:.2f} ms, graph done after {t_all:.2f} ms  {'OK' if good else 'BAD'}")
print("EXTERNAL_EVENT_OVERLAP", "works" if ok else "FAILS")
