from . import build

print(build(force=True))
