from . import build, build_sig

print(build(force=True))
print(build_sig(force=True))
