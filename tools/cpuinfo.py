import os, time, torch
print("cpu_count", os.cpu_count(), "threads", torch.get_num_threads())
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|Flags' | cut -c1-300 | head -8; free -g | head -2")
for dt in (torch.bfloat16, torch.float32):
    a = torch.randn(1024, 3584).to(dt); b = torch.randn(18944, 3584).to(dt)
    torch.nn.functional.linear(a, b)
    t=time.time(); 
    for _ in range(3): torch.nn.functional.linear(a, b)
    dt_s=(time.time()-t)/3
    print(dt, "linear 1024x3584x18944", dt_s, "s", 2*1024*3584*18944/dt_s/1e12, "TFLOP/s")
