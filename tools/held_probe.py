import sys, time, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_cols as T
from vlpet_amd import _lib
lib = _lib.load()
for M in (2000, 9000):
  run, _ = T._abi_case(M)
  ref = run([3], True, True)
  for held in (0, 64, 128, 200, 240, 248):
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream(priority=-1)
    torch.cuda.synchronize()
    if held:
        assert lib.vlpet_test_hold_cus(held, 100 * 1024, flag.data_ptr(), 6000, side.cuda_stream) == 0
        time.sleep(0.05)
    for ph in ([1], [3]):
        t0 = time.time()
        got = run(ph, True, True, sync_device=False)
        t1 = time.time() - t0
        print(f"M {M} held {held} phases {ph}: {t1*1e3:.1f} ms", flush=True)
    t0 = time.time()
    flag.fill_(1)
    torch.cuda.synchronize()
    print(f"   release + sync {1e3*(time.time()-t0):.1f} ms; equal: {all(torch.equal(a.float(), b) for a, b in zip(got, ref))}", flush=True)
