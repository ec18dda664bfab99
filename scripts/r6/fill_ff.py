import torch
xs=[]
for i in range(40):
    xs.append(torch.full((1<<28,), -1, dtype=torch.int64, device="cuda"))   # 2 GB of 0xFF each
torch.cuda.synchronize(); del xs; torch.cuda.empty_cache()
print("filled")
