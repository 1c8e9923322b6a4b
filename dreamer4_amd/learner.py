def learn(*a, **k):
    raise NotImplementedError
