"""Bounded per-shape caches.

The host side keeps resources per batch geometry - captured hipGraphs, pre-allocated noise / solver-state buffers (a PC-100 sampler at
256 clouds holds ~92 MB of them), pinned staging buffers, encoder workspaces (~1.5 MB per cloud).  The reference's evaluation loop
produces one ragged tail per category (evaluation_single.py:381-382), a detector a different object count per image: the number of
distinct shapes is not bounded by the caller, so the caches are - least recently used first.  Dropping an entry drops the last
reference to its graphs and buffers (the caching allocator gets the memory back; a graph's private pool dies with the graph).
"""
from collections import OrderedDict


class ShapeCache:
    def __init__(self, capacity=8, can_evict=None, on_evict=None):
        """can_evict(value) -> bool: entries that must stay (e.g. a workspace a captured graph writes into) are skipped;
        on_evict(key, value): called for every entry dropped."""
        self.capacity, self.can_evict, self.on_evict = int(capacity), can_evict, on_evict
        self._d = OrderedDict()

    def get(self, key, default=None):
        if key in self._d:
            self._d.move_to_end(key)
            return self._d[key]
        return default

    def __contains__(self, key):
        return key in self._d

    def __len__(self):
        return len(self._d)

    def __setitem__(self, key, value):
        old = self._d.get(key, self)
        if old is not self and old is not value and self.on_evict is not None:
            self.on_evict(key, old)  # an overwritten entry is dropped like an evicted one (its pins go back)
        self._d[key] = value
        self._d.move_to_end(key)
        if len(self._d) > self.capacity:
            for k in list(self._d):
                if len(self._d) <= self.capacity:
                    break
                if k == key or (self.can_evict is not None and not self.can_evict(self._d[k])):
                    continue
                v = self._d.pop(k)
                if self.on_evict is not None:
                    self.on_evict(k, v)

    def __getitem__(self, key):
        v = self.get(key, self)
        if v is self:
            raise KeyError(key)
        return v

    def pop(self, key, default=None):
        """Removes and returns the entry; on_evict runs for it (whatever it pinned is released), like for every other way out."""
        if key not in self._d:
            return default
        v = self._d.pop(key)
        if self.on_evict is not None:
            self.on_evict(key, v)
        return v

    def values(self):
        return self._d.values()

    def keys(self):
        return self._d.keys()

    def items(self):
        return self._d.items()

    def clear(self):
        if self.on_evict is not None:
            for k, v in list(self._d.items()):
                self.on_evict(k, v)
        self._d.clear()
