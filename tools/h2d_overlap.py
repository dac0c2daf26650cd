#!/usr/bin/env python
"""Where does the image upload of a pipelined step belong?  bench.py's `pcie_inclusive` leg (the 7.2 MB fp32 image copied host -> device on the MAIN
stream before every pipelined step) comes out 10 % below the headline although the copy is ~0.15 ms of a 4.5 ms step.  Same steps, four placements of
the copy.  GPU only; prints img/s per form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100


class H(bench.Hooks):
    def after_optim(self, optim):
        H.optim = optim

    def after_net(self, net):
        H.net = net


def main():
    # build the network / optimiser / blobs exactly as bench does, with a short headline run
    bench.main(['--steps', '5', '--warmup', '3', '--no-cpu-baseline', '--extras', '0', '--mixed-shapes', '0'], H())
    net, optim = H.net, H.optim
    from lang2seg_amd.loaders.synthetic_loader import SyntheticLoader
    loader = SyntheticLoader(num_images=4, sents_per_image=1, H=600, W=1000, T=20, vocab_size=3349)
    blobs = [loader.getBatch('train') for _ in range(4)]
    for b in blobs:
        net.upload_blob(b, 0)
    hosts = [torch.from_numpy(np.ascontiguousarray(b['data'], dtype=np.float32)).pin_memory() for b in blobs]
    devd = [b['_device']['data'] for b in blobs]
    main_s = torch.cuda.current_stream()
    cp = torch.cuda.Stream()

    def run(form):
        for i in range(8):
            net.train_step_async(blobs[i % 4], 0, optim)
        torch.cuda.synchronize()
        ev, rd = [torch.cuda.Event() for _ in range(4)], [None] * 4
        if form == 'prefetch':                                   # the first image is there before the clock starts
            with torch.cuda.stream(cp):
                devd[0].copy_(hosts[0], non_blocking=True); ev[0].record(cp)
        t0 = time.time()
        for i in range(STEPS):
            j = i % 4
            if form == 'main':
                devd[j].copy_(hosts[j], non_blocking=True)
            elif form == 'side':                                 # the copy on its own stream, ordered behind the previous users of the buffer, main waits for it
                cp.wait_stream(main_s)
                with torch.cuda.stream(cp):
                    devd[j].copy_(hosts[j], non_blocking=True)
                main_s.wait_stream(cp)
            elif form == 'prefetch':                             # step i waits for ITS image; the NEXT image is uploaded beside step i
                main_s.wait_event(ev[j])
            net.train_step_async(blobs[j], 0, optim)
            if form == 'prefetch':
                jn = (i + 1) % 4
                rd[j] = torch.cuda.Event(); rd[j].record(main_s)  # behind the step that read buffer j
                if rd[jn] is not None:
                    cp.wait_event(rd[jn])                         # the upload may not overtake the last reader of the buffer it overwrites
                with torch.cuda.stream(cp):
                    devd[jn].copy_(hosts[jn], non_blocking=True); ev[jn].record(cp)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print('%-10s %7.2f img/s  %.3f ms' % (form, STEPS / dt, dt / STEPS * 1e3), flush=True)
    for rep in range(2):
        for form in ('none', 'main', 'side', 'prefetch'):
            run(form)


if __name__ == '__main__':
    main()
