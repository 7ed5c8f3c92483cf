def pytest_configure(config):
    config.addinivalue_line("markers", "experimental: kernels of experiments/ (not part of the product library)")
    config.addinivalue_line("markers", "gpu: needs a real MI355X")
