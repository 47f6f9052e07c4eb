class DataLoader:  # import-only on the hot path
    def __init__(self, *a, **k):
        raise NotImplementedError


class DataListLoader(DataLoader):
    pass
