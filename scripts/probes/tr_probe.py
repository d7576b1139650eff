"""ds_read_b64_tr_b16 semantics: LDS u16[i] = i.  Case A: lane l reads 8 bytes at row-major address of a [16 rows x 64 B] tile
(row = l % 16, 8-byte chunk = l // 16).  Prints what each lane receives."""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libtrprobe.so"))
dev = torch.device("cuda:0")
def run(addr):
    a = torch.tensor(addr, dtype=torch.int32, device=dev)
    out = torch.zeros(64 * 4, dtype=torch.int16, device=dev)
    rc = lib.run_tr(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 4096, None)
    torch.cuda.synchronize(); assert rc == 0
    return out.cpu().view(64, 4).tolist()
RS = 64  # row stride in u16 elements (128 B rows)
print("case A: lane l -> row l%16, chunk l//16 (elements row*64 + 4*chunk .. +3)")
res = run([((l % 16) * RS + 4 * (l // 16)) * 2 for l in range(64)])
for l in range(64):
    print(l, [(v // RS, v % RS) for v in res[l]], end=" | " if l % 4 != 3 else "\n")
print("case B: lane l -> row l//4 ... (16 rows, 4 chunks per row by l%4)")
res = run([((l // 4) * RS + 4 * (l % 4)) * 2 for l in range(64)])
for l in range(64):
    print(l, [(v // RS, v % RS) for v in res[l]], end=" | " if l % 4 != 3 else "\n")
