"""Attribute bag + the collate rules of PyG's Data/Batch (restated)."""
import torch


class Data:
    def __init__(self, x=None, edge_index=None, **kwargs):
        self.__dict__['_store'] = {}
        if x is not None:
            self.x = x
        if edge_index is not None:
            self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __setattr__(self, key, value):
        if key.startswith('_'):
            self.__dict__[key] = value
        else:
            self._store[key] = value

    def __getattr__(self, key):
        store = self.__dict__.get('_store', {})
        if key in store:
            return store[key]
        raise AttributeError(key)

    def __getitem__(self, key):
        return self._store[key]

    def __setitem__(self, key, value):
        self._store[key] = value

    def __contains__(self, key):
        return key in self._store

    @property
    def keys(self):
        return [k for k, v in self._store.items() if v is not None]

    @property
    def num_nodes(self):
        return self._store['x'].size(0)

    def __cat_dim__(self, key, value, *args, **kwargs):
        return -1 if 'index' in key else 0

    def __inc__(self, key, value, *args, **kwargs):
        return self.num_nodes if 'index' in key else 0

    def to(self, device):
        for k, v in self._store.items():
            if torch.is_tensor(v):
                self._store[k] = v.to(device)
        return self


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list):
        """PyG collate: cat along __cat_dim__ (None -> stack), add the running
        sum of __inc__ to every item, and emit `batch` (graph id per node)."""
        out = cls()
        keys = data_list[0].keys
        for key in keys:
            items, inc = [], 0
            for d in data_list:
                v = d[key]
                if torch.is_tensor(v):
                    shifted = torch.is_tensor(inc) or inc != 0
                    items.append(v + inc if shifted else v)
                    inc = inc + d.__inc__(key, v)
                else:
                    items.append(v)
            v0 = data_list[0][key]
            if torch.is_tensor(v0):
                cat_dim = data_list[0].__cat_dim__(key, v0)
                if cat_dim is None:
                    out[key] = torch.stack(items, 0)
                else:
                    out[key] = torch.cat(items, cat_dim)
            else:
                out[key] = items
        out['batch'] = torch.cat([torch.full((d.num_nodes,), i, dtype=torch.long)
                                  for i, d in enumerate(data_list)])
        out._num_graphs = len(data_list)
        return out


class DataLoader:  # import-only; the hot path never constructs one
    def __init__(self, *a, **k):
        raise NotImplementedError


class DataListLoader(DataLoader):
    pass
