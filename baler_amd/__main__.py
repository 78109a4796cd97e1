from .baler import main

main()
