import ctypes, os
l = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libt.so"))
print(l.run(3), l.run(5))
