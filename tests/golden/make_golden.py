"""Generate tests/golden/ref_kernels.npz from the reference's own OpenCL C kernels.

Run in the build container, where /root/reference exists:
    make -C oracle ref && python tests/golden/make_golden.py
The kernels are compiled by oracle/Makefile from where they lie under
/root/reference/modules (rndgenmwc64x/cl/randstategen.cl + random.cl + skip_mwc.cl,
progressivephotonmapping/cl/{densityestimationkernel,threshold,indextobuffer,photon}.cl, uniformgridcl/cl/buffermixer.cl,
rndgenmwc64x/cl/randomnumbergenerator.cl) and driven by oracle/ref_harness.c, oracle/ref_harness_vec.c and oracle/ref_harness2.c.  The fixture holds inputs and the outputs those kernels produced --
data only; no reference source travels.
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle_binding import Ref  # noqa: E402

OUT = Path(__file__).resolve().parent / "ref_kernels.npz"


def main():
    ref = Ref()
    n = 4096
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(0)
    bases = np.array([libc.rand() for _ in range(n)], dtype=np.uint32)  # what mwc64xseedgenerator.cpp:56-64 draws

    state = np.zeros((n, 2), np.uint32)
    state[:, 0] = bases
    ref.generate_random_state(state)                      # MWC64X_GenerateRandomState, gap 2^40
    seeded = state.copy()
    r01, ruint = ref.random_fill(state, 8)                # random_01 / MWC64X_NextUint
    state_after = state.copy()

    ps = np.zeros((257, 2), np.uint32)
    ps[:, 0] = bases[:257]
    ref.generate_per_stream_random_state(ps, 1000)        # MWC64X_GeneratePerStreamRandomState

    rng = np.random.default_rng(12345)
    kx = np.concatenate([np.linspace(0, 1.5, 2049, dtype=np.float32),
                         rng.random(2048, dtype=np.float32),
                         np.array([1.0, np.nextafter(np.float32(1), np.float32(2)), np.nextafter(np.float32(1), np.float32(0))], np.float32)])
    ky = ref.density_kernel(kx)                           # densityEstimationKernel

    tdata = rng.integers(0, 2**32, 1000, dtype=np.uint64).astype(np.uint32)
    tdata[::7] = 2147483647
    tdata[::11] = 2147483646
    tout = ref.threshold(tdata, 2147483647)               # thresholdKernel
    iota = ref.index_to_buffer(777)                       # indexToBufferKernel

    # mixKernel (uniformgridcl/cl/buffermixer.cl) as built for float grids and for the min/max grid
    mx = (rng.standard_normal(1003) * 7).astype(np.float32)
    my = (rng.standard_normal(1003) * 7).astype(np.float32)
    mix_a = np.array([0.0, 0.25, 0.37, 0.999, 1.0], np.float32)
    mix_f = np.stack([ref.mix_f32(mx, my, float(a)) for a in mix_a])
    ux = rng.integers(0, 65536, (517, 2)).astype(np.uint16)
    uy = rng.integers(0, 65536, (517, 2)).astype(np.uint16)
    ux[0], uy[0] = (0, 65535), (65535, 0)
    mix_u = np.stack([ref.mix_u16x2(ux, uy, float(a)) for a in mix_a])

    # randomNumberGeneratorKernel (rndgenmwc64x/cl/randomnumbergenerator.cl): loadRandState -> random_01 -> saveRandState,
    # three launches in a row over 1000 streams (not a multiple of the launch's rounding)
    ks = seeded[:1000].copy()
    k_draws = np.stack([ref.random_number_kernel(ks) for _ in range(3)])
    k_state = ks.copy()
    # readPhoton / writePhoton (progressivephotonmapping/cl/photon.cl): 300 records written at shuffled ids of a 512-record buffer
    ph = (rng.standard_normal((300, 8)) * 3).astype(np.float32)
    ph[5, :3] = np.float32(3.402823466e+38)
    ids = rng.permutation(512)[:300].astype(np.int32)
    ph_buf, ph_back = ref.photon_write_read(ph, ids, 512)

    np.savez_compressed(OUT, rng_kernel_draws=k_draws, rng_kernel_state=k_state, photon_in=ph, photon_ids=ids, photon_buffer=ph_buf,
                        photon_read_back=ph_back, mix_x=mx, mix_y=my, mix_a=mix_a, mix_f32=mix_f, mix_ux=ux, mix_uy=uy, mix_u16x2=mix_u, bases=bases, seeded=seeded, random01=r01, random_uint=ruint, state_after=state_after,
                        per_stream_seeded=ps, per_stream_gap=np.uint64(1000), kernel_x=kx, kernel_y=ky,
                        threshold_in=tdata, threshold_out=tout, iota=iota)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
