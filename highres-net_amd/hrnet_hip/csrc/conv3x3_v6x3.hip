// conv3x3 in split-bf16 ("bf16x3"): the precision mode that meets the reference's fp32 arithmetic (train.py:168-171, predict.py:36-37)
// to ~1e-5 at matrix-core speed.  Every fp32 value v travels as two bf16 planes, hi = bf16(v) and lo = bf16(v - hi) (~16 significant
// bits), weights likewise (packed as two planes), and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation (the dropped lo*lo term is ~2^-16 of the product).  The kernel is conv3x3_v6's (conv3x3_v6_impl.h) with the K loop run
// three passes per chunk and an epilogue that splits the fp32 result again; here are its instantiations and their launcher:
//   64 -> 64   (+ plain residual)                 the encoder's five layers                       HRNet.py:17-22, :55-60
//   128 -> 128 (pair gather in / residual)        the fusion ResidualBlock                        HRNet.py:90-94, :113-119
//   128 -> 64  (+ alpha residual into the stack)  the fusion output conv                          HRNet.py:95-97, :123-131
//   64 -> 128, 128 -> 128 + plain residual        data gradients of the training path (a cout -> cin convolution on transposed weights)
#include "conv3x3_v6_impl.h"

int hrn_launch_conv3x3_v6x3(int cin, int cout, const ConvParams& p, hipStream_t stream) {
    HRN_CHECK(!p.scale && !p.relu, -2, "conv3x3 bf16x3: folded scale / ReLU are not supported");
    const bool ok = (cin == 64 && cout == 64 && !p.in_pair && (p.res_mode == 0 || p.res_mode == 1)) ||
                    (cin == 64 && cout == 128 && !p.in_pair && p.res_mode == 0) ||
                    (cin == 128 && cout == 128 && (p.res_mode == 0 || p.res_mode == 2 || (p.res_mode == 1 && !p.in_pair))) ||
                    (cin == 128 && cout == 64 && !p.in_pair && (p.res_mode == 0 || p.res_mode == 3));
    HRN_CHECK(ok, -2, "conv3x3 bf16x3: unsupported layer cin=%d cout=%d res_mode=%d in_pair=%d", cin, cout, p.res_mode, p.in_pair);
    HRN_CHECK(!((p.in_pair || p.res_mode == 2) && p.pair_h <= 0), -2, "conv3x3 bf16x3: pair descriptor missing");
    HRN_CHECK(!(p.res_mode == 3 && (p.out_h <= 0 || !p.res)), -2, "conv3x3 bf16x3: res_mode 3 needs slot output and a residual");
    HRN_CHECK(!(p.res_mode == 1 && !p.res), -2, "conv3x3 bf16x3: res_mode 1 needs a residual tensor");
    HRN_CHECK(p.out_lo != 0 && (p.in_pair ? p.stack_lo != 0 : p.in_lo != 0), -2, "conv3x3 bf16x3: lo-plane offsets missing");
    long grid = 0;
    const int rc = v6_grid(p, cin, grid);
    HRN_CHECK(rc != -100, -2, "conv3x3 bf16x3: image too large for 32-bit in-image offsets (H=%d W=%d)", p.H, p.W);
    if (rc) return rc;
    const double px = (double)p.M * p.H * p.W;
    static const char* fams[4][2] = {{"conv3x3_bf16x3_64x64", "conv3x3_bf16x3_64x64+res"}, {"conv3x3_bf16x3_128x128", "conv3x3_bf16x3_128x128+res"},
                                     {"conv3x3_bf16x3_128x64", "conv3x3_bf16x3_128x64+res"}, {"conv3x3_bf16x3_64x128", "conv3x3_bf16x3_64x128+res"}};
    const int fi = cin == 64 ? (cout == 64 ? 0 : 3) : (cout == 128 ? 1 : 2);
    // algorithmic FLOPs of the layer (not x 3) and bytes of both planes
    HrnProfScope prof(fams[fi][p.res_mode ? 1 : 0], 2.0 * cin * cout * 9 * px, px * 4 * (cin + cout + (p.res_mode ? cout : 0)), stream);
    if (cin == 64 && cout == 128) return launch_v6<64, 128, 0, false, true>(p, grid, stream);      // the data gradient of a 128 -> 64 layer
    if (cin == 64) return p.res_mode ? launch_v6<64, 64, 1, false, true>(p, grid, stream) : launch_v6<64, 64, 0, false, true>(p, grid, stream);
    if (cout == 128) {
        if (p.in_pair) return p.res_mode ? launch_v6<128, 128, 2, true, true>(p, grid, stream) : launch_v6<128, 128, 0, true, true>(p, grid, stream);
        if (p.res_mode == 1) return launch_v6<128, 128, 1, false, true>(p, grid, stream);                  // data gradient + the gradient that bypasses the layer
        return p.res_mode ? launch_v6<128, 128, 2, false, true>(p, grid, stream) : launch_v6<128, 128, 0, false, true>(p, grid, stream);
    }
    return p.res_mode ? launch_v6<128, 64, 3, false, true>(p, grid, stream) : launch_v6<128, 64, 0, false, true>(p, grid, stream);
}
