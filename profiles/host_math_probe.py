"""Which bits does the HOST's torch CPU produce for exp/log/pow/atan2?  (MKL VML dispatches on the CPU type.)
Prints sha256 of the outputs on fixed inputs so two hosts can be compared."""
import hashlib, subprocess, torch
print(subprocess.run("lscpu | grep -E 'Model name|Flags' | cut -c1-400", shell=True, capture_output=True, text=True).stdout)
print(torch.__version__, torch.backends.cpu.get_cpu_capability())
g = torch.Generator().manual_seed(1234)
bits = torch.randint(0, 2**31 - 1, (1 << 22,), dtype=torch.int32, generator=g)
x = bits.view(torch.float32)
x = x[(x.abs() < 80) & (x.abs() > 1e-6)]
x = torch.cat([x, -x]).contiguous()
h = lambda t: hashlib.sha256(t.contiguous().view(torch.int32).numpy().tobytes()).hexdigest()[:16]
print("exp", h(torch.exp(x)))
print("log", h(torch.log(x.abs())))
u = torch.rand(1 << 22, generator=g)
print("pow", h(u.pow(1 / 2.33)))
a = torch.randn(1 << 22, generator=g); b = torch.randn(1 << 22, generator=g)
print("atan2", h(torch.atan2(a, b)))
print("lse", h(torch.logsumexp(torch.randn(1 << 18, 4, 1, generator=g) * 5, dim=-2)))
