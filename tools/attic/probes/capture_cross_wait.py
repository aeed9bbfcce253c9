"""What stream-capture patterns does this ROCm survive?  (hipStreamEndCapture segfaults on some event topologies.)
0: fork / join through the origin stream; 1: two side streams wait on each other's events, kernels after; 2: the same, no kernels after;
3: one side stream waits on the other's event (one direction); 4: join immediately followed by a fork, no kernel on the origin in between."""
import torch, sys
dev = torch.device('cuda:0')
def run(pattern):
    main = torch.cuda.Stream(dev); a = torch.cuda.Stream(dev); b = torch.cuda.Stream(dev)
    x = torch.zeros(1 << 16, device=dev); y = torch.zeros(1 << 16, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        e = torch.cuda.Event(); e.record(main); a.wait_event(e); b.wait_event(e)
        with torch.cuda.stream(a): x.add_(1)
        with torch.cuda.stream(b): y.add_(1)
        if pattern in (1, 2):
            ea = torch.cuda.Event(); eb = torch.cuda.Event(); ea.record(a); eb.record(b)
            a.wait_event(eb); b.wait_event(ea)
            if pattern == 1:
                with torch.cuda.stream(a): x.add_(1)
                with torch.cuda.stream(b): y.add_(1)
        if pattern == 3:
            eb = torch.cuda.Event(); eb.record(b); a.wait_event(eb)
            with torch.cuda.stream(a): x.add_(1)
            with torch.cuda.stream(b): y.add_(1)
        if pattern == 4:
            ea2 = torch.cuda.Event(); eb2 = torch.cuda.Event(); ea2.record(a); eb2.record(b)
            main.wait_event(ea2); main.wait_event(eb2)
            e = torch.cuda.Event(); e.record(main); a.wait_event(e); b.wait_event(e)
            with torch.cuda.stream(a): x.add_(1)
            with torch.cuda.stream(b): y.add_(1)
        ea2 = torch.cuda.Event(); eb2 = torch.cuda.Event(); ea2.record(a); eb2.record(b)
        main.wait_event(ea2); main.wait_event(eb2)
        x.add_(1)
    g.replay(); torch.cuda.synchronize()
    print('pattern', pattern, 'ok', x[0].item(), y[0].item(), flush=True)
run(int(sys.argv[1]))
