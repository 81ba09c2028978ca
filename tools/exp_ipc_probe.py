"""Can two processes on one MI355X share device memory and order work across the process boundary?  (probe for a helper
process that runs the frozen teacher: tensors through torch.multiprocessing's IPC reductions, an inter-process event.)"""
import os, sys, time
import torch
import torch.multiprocessing as mp


def child(q_in, q_out):
    torch.cuda.set_device(0)
    t = q_in.get()                      # a tensor that lives in the parent
    t.add_(5)
    torch.cuda.synchronize()
    q_out.put('added')
    # a tensor that lives in the child
    own = torch.arange(8, device='cuda', dtype=torch.float32) * 2
    ev = None
    try:
        ev = torch.cuda.Event(interprocess=True)
        torch.cuda._sleep(50_000_000)
        own.mul_(3)
        ev.record()
        q_out.put(('own', own, ev.ipc_handle()))
    except Exception as e:              # noqa: BLE001
        torch.cuda.synchronize()
        q_out.put(('own', own, None))
        q_out.put('event failed: %r' % (e,))
    q_in.get()                          # keep `own` alive until the parent is done


if __name__ == '__main__':
    ctx = mp.get_context('spawn')
    q_in, q_out = ctx.Queue(), ctx.Queue()
    p = ctx.Process(target=child, args=(q_in, q_out))
    p.start()
    x = torch.ones(4, device='cuda')
    t0 = time.perf_counter()
    q_in.put(x)
    print(q_out.get(timeout=120), 'parent sees', x.tolist(), 'in %.1f ms' % ((time.perf_counter() - t0) * 1e3))
    msg = q_out.get(timeout=120)
    _, own, handle = msg
    if handle is not None:
        ev = torch.cuda.Event.from_ipc_handle(torch.device('cuda', 0), handle)
        torch.cuda.current_stream().wait_event(ev)
        y = own + 1
        print('child tensor behind its event:', y.tolist())
    else:
        print(q_out.get(timeout=10))
        print('child tensor (host handshake):', (own + 1).tolist())
    # cost of receiving a tensor per step
    q_in.put('done')
    p.join(30)
    print('child exit', p.exitcode)
