"""Import shim (build container only): no-op SummaryWriter (only constructed when cfg.is_train)."""
class SummaryWriter:
    def __init__(self, *a, **k): pass
    def __getattr__(self, name):
        return lambda *a, **k: None
