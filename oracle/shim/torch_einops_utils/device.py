def move_inputs_to_module_device(fn): return fn
