"""A/B two builds of liblasr.so op by op on identical inputs (dev tool)."""
import sys, os, ctypes as C
sys.path.insert(0, '.')
import torch
from lightning_asr_amd import _lib, ops

def load(path):
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res; fn.argtypes = args
    return lib

def main():
    new = load(os.path.abspath('lightning_asr_amd/liblasr.so'))
    old = load(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else 'build/liblasr_old.so'))
    dev = torch.device('cuda')

    def both(fn):
        out = []
        for lib in (old, new):
            _lib._lib = lib
            out.append(fn())
        return out

    def rel(a, b):
        a, b = a.double(), b.double()
        return ((a - b).norm() / (b.norm() + 1e-30)).item()

    g = torch.Generator().manual_seed(0)
    B, T = 4, 101
    for C_, has_res, masked in [(1024, False, False), (512, True, True), (256, True, True)]:
        y = torch.randn(B, T, C_, generator=g).to(dev); y2 = torch.randn(B, T, C_, generator=g).to(dev) if has_res else None
        dout = torch.randn(B, T, C_, generator=g).to(dev)
        gam = (1 + 0.1 * torch.randn(C_, generator=g)).to(dev); bet = (0.1 * torch.randn(C_, generator=g)).to(dev)
        lens = torch.tensor([101, 101, 90, 50], dtype=torch.int32, device=dev)
        N = B * T
        def run():
            st = torch.cat([y.reshape(N, C_).sum(0), (y.reshape(N, C_) ** 2).sum(0)])
            coef, saved = ops.bn_finalize(st, gam, bet, torch.zeros(C_, device=dev), torch.ones(C_, device=dev), N)
            coef2 = saved2 = None
            if has_res:
                st2 = torch.cat([y2.reshape(N, C_).sum(0), (y2.reshape(N, C_) ** 2).sum(0)])
                coef2, saved2 = ops.bn_finalize(st2, gam, bet, torch.zeros(C_, device=dev), torch.ones(C_, device=dev), N)
            o = ops.bn_act(y, coef, y2, coef2, None, "relu")
            r = ops.bn_act_bwd(dout, y, coef, saved, gam, y2, coef2, saved2, gam if has_res else None, row_lens=lens if masked else None)
            return [o] + [t for t in r if t is not None]
        a, b = both(run)
        print("bn C=%d res=%d" % (C_, has_res), ["%.1e" % rel(p, q) for p, q in zip(b, a)])

    for C_, k in [(512, 63), (256, 33), (512, 75)]:
        x = torch.randn(B, T, C_, generator=g).to(dev); w = (torch.randn(C_, k, generator=g) / 8).to(dev)
        add = torch.randn(B, T, C_, generator=g).to(dev)
        a, b = both(lambda: [ops.dwconv(x, w), ops.dwconv(x, w, flip=True, addend=add), ops.dwconv_wgrad(x, add, k)])
        print("dw C=%d k=%d" % (C_, k), ["%.1e" % rel(p, q) for p, q in zip(b, a)])

    lp = torch.log_softmax(torch.randn(B, T, 28, generator=g) * 2, -1).to(dev)
    tg = torch.randint(0, 27, (B, 12), generator=g).to(dev)
    il = torch.tensor([101, 101, 90, 50], dtype=torch.int32, device=dev); tl = torch.tensor([12, 9, 7, 5], dtype=torch.int32, device=dev)
    a, b = both(lambda: list(ops.ctc_loss(lp, tg, il, tl, 27)))
    print("ctc", ["%.1e" % rel(p, q) for p, q in zip(b, a)])
    A = torch.randn(404, 512, generator=g).to(dev); W = torch.randn(1024, 512, generator=g).to(dev)
    a, b = both(lambda: list(ops.gemm(A, W, 404, 1024, 512, want_stats=True)))
    print("gemm stats", ["%.1e" % rel(p, q) for p, q in zip(b, a)])


if __name__ == '__main__':
    main()
