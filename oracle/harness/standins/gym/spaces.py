import random
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype


class Discrete(Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)

    def sample(self):
        return self.start + random.randrange(self.n)

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n

    def __repr__(self):
        return f"Discrete({self.n})"


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        super().__init__(tuple(shape) if shape is not None else np.shape(low), dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))
