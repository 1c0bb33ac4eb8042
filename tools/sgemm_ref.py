"""What the vendor library reaches on the GEMM shapes behind the batch ResBlock convs (fp32, exact): torch.matmul (rocBLAS / hipBLASLt) on
[C_out x C_in*K] x [C_in*K x columns] -- the implicit GEMM of conv_mfma_kernel WITHOUT its im2col (the library gets the unfolded operand
for free).  A yardstick for "how close to the 157.3 TFLOP/s fp32 MFMA peak does a tuned kernel get on this shape", not a competitor
(materialising the unfolded operand would cost K x the activation traffic).      python tools/sgemm_ref.py"""
import time, torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda", 0)
print(torch.__version__, torch.cuda.get_device_name(0))
for (M, Kc, N, what) in [(128, 128 * 11, 194688, "stage 2, k=11: C=128, 12168 frames x 16"), (128, 128 * 3, 194688, "stage 2, k=3"),
                         (256, 256 * 11, 48672, "stage 1, k=11: C=256, 12168 frames x 4"), (256, 256 * 7, 48672, "stage 1, k=7"), (256, 256 * 3, 48672, "stage 1, k=3"),
                         (4096, 4096, 4096, "square 4096 (the library's home ground)"), (8192, 8192, 8192, "square 8192")]:
    a = torch.randn(M, Kc, device=dev) * 0.05
    b = torch.randn(Kc, N, device=dev) * 0.5
    for _ in range(3): c = a @ b
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * Kc * N / (ms * 1e-3) / 1e12
    print(f"{what:45s} M={M:5d} K={Kc:5d} N={N:6d}: {ms*1e3:8.1f} us  {tf:6.1f} TFLOP/s = {tf/157.3:.3f} of the fp32 MFMA peak")
