def save_load(*a, **k):
    if len(a) == 1 and isinstance(a[0], type): return a[0]
    return lambda klass: klass
