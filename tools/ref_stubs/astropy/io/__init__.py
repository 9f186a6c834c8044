fits = None
ascii = None
