/* sigmaenv_ref_weights.h -- weighting_ref_directions = torch.linspace(1, 0.2, n_points_short_term) / sum (road_traffic.py:536-543) as the bit patterns of
 * the reference's own float32 tensor for n_points_short_term = 1 .. 8 (read from PyTorch-CPU in the build container: tools/ref_weights.py prints this
 * table; a restated linspace could differ in the last bit).  Shared by the HIP kernels and the CPU oracle. */
#ifndef SIGMAENV_REF_WEIGHTS_H
#define SIGMAENV_REF_WEIGHTS_H
#if SIGMAENV_N_SHORT_TERM == 1
#define SIGMAENV_W_REF_BITS {0x3F800000u}
#elif SIGMAENV_N_SHORT_TERM == 2
#define SIGMAENV_W_REF_BITS {0x3F555555u, 0x3E2AAAAAu}
#elif SIGMAENV_N_SHORT_TERM == 3
#define SIGMAENV_W_REF_BITS {0x3F0E38E3u, 0x3EAAAAABu, 0x3DE38E39u}
#elif SIGMAENV_N_SHORT_TERM == 4
#define SIGMAENV_W_REF_BITS {0x3ED55555u, 0x3E9C71C7u, 0x3E471C72u, 0x3DAAAAAAu}
#elif SIGMAENV_N_SHORT_TERM == 5
#define SIGMAENV_W_REF_BITS {0x3EAAAAABu, 0x3E888889u, 0x3E4CCCCDu, 0x3E088889u, 0x3D888889u}
#elif SIGMAENV_N_SHORT_TERM == 6
#define SIGMAENV_W_REF_BITS {0x3E8E38E3u, 0x3E6EEEEFu, 0x3E416C16u, 0x3E13E93Eu, 0x3DCCCCCDu, 0x3D638E39u}
#elif SIGMAENV_N_SHORT_TERM == 7
#define SIGMAENV_W_REF_BITS {0x3E73CF3Cu, 0x3E534D34u, 0x3E32CB2Cu, 0x3E124924u, 0x3DE38E39u, 0x3DA28A28u, 0x3D430C30u}
#elif SIGMAENV_N_SHORT_TERM == 8
#define SIGMAENV_W_REF_BITS {0x3E555556u, 0x3E3CF3D0u, 0x3E24924Au, 0x3E0C30C4u, 0x3DE79E7Cu, 0x3DB6DB6Fu, 0x3D861862u, 0x3D2AAAACu}
#else
#error "SIGMAENV_N_SHORT_TERM must be 1 .. 8"
#endif
#endif
