import os

strides = [8, 16, 32, 64, 128]
levels = len(strides)
tag = "s" + str(strides[0])
out = os.path.join(tag, "ckpt")
